NAME reappear
ROWS
 N obj
 L r1
 L r2
COLUMNS
 x obj 1 r1 1
 y obj 2 r1 1
 x obj 3 r2 1
 y r2 4
BOUNDS
 UP b x 7
 UP b y 8
ENDATA
