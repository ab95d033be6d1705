NAME numbers
ROWS
 N obj
 L r1
 G r2
COLUMNS
	x	obj	1.0D3	r1	+2.5
 x r2 1.5d-2
 y obj .5 r1 5.
 y r2 -0.0
 z obj 1e19 r1 1E-320
 z r2 123456789.123456789e-3
 w obj 0x10 r1 1.5abc
 w r2 +.25e+1
 v obj 7E0 r1 0.1
 v r2 1e-9
RHS
 rhs r1 1e30 r2 -1E+20
BOUNDS
 UP b x 1e20
 LO b y -Inf
 UP b z Infinity
 LO b w -1D1
ENDATA