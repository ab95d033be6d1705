NAME qs
ROWS
 N obj
 G c1
 N other
COLUMNS
 x obj 1 c1 1
 y obj 1 c1 1
RHS
 rhs c1 1
QSECTION other
 x x 9.0
QSECTION obj
 x x 2.0 y 0.25
 y y 1.0
QSECTION undefinedrow
 x x 5.0
ENDATA
