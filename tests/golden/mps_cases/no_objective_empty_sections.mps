NAME noobj
ROWS
 L r1
 G r2
COLUMNS
 x r1 1 r2 1
 y r1 1
RHS
RANGES
BOUNDS
ENDATA
junk after the end
