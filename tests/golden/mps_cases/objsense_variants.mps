NAME sense1
OBJSENSE
    MAXIMIZE
ROWS
 N obj
 L r
COLUMNS
 x obj 1 r 1
RHS
 rhs r 4
ENDATA
