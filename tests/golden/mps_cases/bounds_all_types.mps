NAME bounds
ROWS
 N obj
 L c1
 G c2
COLUMNS
 a obj 1 c1 1
 b obj 1 c1 1
 c obj 1 c1 1
 d obj 1 c2 1
 e obj 1 c2 1
 f obj 1 c2 1
 g obj 1 c1 1
 h obj 1 c1 1
 i obj 1 c2 1
 j obj 1 c2 1
 k obj 1 c2 1
 l obj 1 c2 1
RHS
 r c1 10 c2 1
BOUNDS
 UP bnd a 4.5
 UP bnd b -2
 LO bnd c 1.25
 FX bnd d 3
 FR bnd e
 MI bnd f
 PL bnd g
 BV bnd h
 LI bnd i 2
 UI bnd i 7
 SC bnd j 10
 SI bnd k 6
 UP bnd a 9
 UP bnd l 1e30
 UP bnd newcol 5
 MI newcol2
 LO a -1e25
ENDATA
