NAME qo
ROWS
 N obj
 G c1
COLUMNS
 x obj 1 c1 1
 y obj 1 c1 1
 z obj 1 c1 1
RHS
 rhs c1 1
QUADOBJ
 x x 2.0 y x 0.5
 y y 3.0
 z y -1.0 z z 4.0
ENDATA
