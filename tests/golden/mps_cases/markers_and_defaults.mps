NAME ints
ROWS
 N obj
 L r1
 L r2
COLUMNS
 c0 obj 1 r1 1
 MARKER 'MARKER' 'INTORG'
 i1 obj 1 r1 1
 i2 obj 1 r2 1
 i3 obj 1 r2 1
 i4 obj 1 r1 1
 MARKER 'MARKER' 'INTEND'
 c5 obj 1 r2 1
 M2 'MARKER' 'INTORG'
 i6 r1 1
 M2 'MARKER' 'INTEND'
RHS
 rhs r1 10 r2 10
BOUNDS
 UP b i2 8
 LO b i3 2
 MI b i4
ENDATA
