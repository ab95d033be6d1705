* RANGES on every row type and sign (HMpsFF.cpp:1554-1566); second pair on a line; undefined and N rows ignored
NAME ranges
ROWS
 N cost
 E e_pos
 E e_neg
 E e_zero
 L l_row
 G g_row
 N free1
COLUMNS
 x cost 1.0 e_pos 1.0
 x e_neg 1.0 e_zero 1.0
 x l_row 2.0 g_row 3.0
 y cost -1.5 l_row 1.0
 y free1 9.0 g_row 1.0
RHS
 rhs e_pos 4.0 e_neg 5.0
 rhs e_zero 6.0 l_row 7.0
 rhs g_row 8.0 cost -2.5
RANGES
 rng e_pos 2.0 e_neg -3.0
 rng e_zero 0.0 l_row -1.5
 rng g_row 2.5 nosuchrow 1.0
 rng cost 1.0 free1 2.0
 rng e_pos 99.0
ENDATA
