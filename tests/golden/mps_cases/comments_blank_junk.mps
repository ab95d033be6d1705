* leading comment

this line is outside any section and is ignored
NAME          junk      extra words
* comment
ROWS
* comment inside
 N  obj

 L  r1
COLUMNS
    x         obj            1.0   r1             1.0
* c
    y         obj            2.0
    y         r1             3.0
RHS
    rhs       r1            10.0
ENDATA
