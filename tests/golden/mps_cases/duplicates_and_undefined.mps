NAME dups
ROWS
 N obj
 N obj2
 L r1
 G r2
 E r1
COLUMNS
 x r1 1.0 r2 2.0
 x r1 5.0 obj 3.0
 x obj 4.0 nosuch 1.0
 x obj2 7.0 r2 0.0
 y r2 0 r1 2.5
 y r2 1.5
 z obj 0.0 r1 1e-12
 x r2 9.0
RHS
 rhs r1 3.0 r1 4.0
 rhs r2 1.0 obj 5.0
 rhs obj 6.0 nosuch 2.0
BOUNDS
 UP b x 4
 UP b x 5
 LO b x 1
 FX b x 2
ENDATA
