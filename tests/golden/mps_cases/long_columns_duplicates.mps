NAME longcols
ROWS
 N obj
 E r0
 G r1
 E r2
 G r3
 E r4
 E r5
 E r6
 E r7
 L r8
 G r9
 L r10
 E r11
 L r12
 L r13
 L r14
 G r15
 G r16
 L r17
 G r18
 E r19
 L r20
 E r21
 L r22
 L r23
 E r24
 L r25
 G r26
 G r27
 L r28
 G r29
 L r30
 L r31
 L r32
 E r33
 E r34
 G r35
 L r36
 L r37
 L r38
 L r39
 L r40
 L r41
 L r42
 L r43
 G r44
 G r45
 L r46
 E r47
 E r48
 E r49
 L r50
 L r51
 E r52
 L r53
 G r54
 G r55
 L r56
 G r57
 G r58
 L r59
 L r60
 G r61
 L r62
 G r63
 G r64
 E r65
 E r66
 L r67
 E r68
 E r69
 E r70
 G r71
 L r72
 G r73
 G r74
 G r75
 G r76
 E r77
 G r78
 L r79
 G r80
 G r81
 E r82
 L r83
 L r84
 G r85
 L r86
 E r87
 G r88
 G r89
 L r90
 E r91
 G r92
 G r93
 G r94
 E r95
 L r96
 G r97
 L r98
 E r99
 L r100
 E r101
 L r102
 L r103
 L r104
 G r105
 G r106
 E r107
 G r108
 E r109
 G r110
 G r111
 L r112
 E r113
 E r114
 G r115
 G r116
 L r117
 G r118
 L r119
 L r120
 G r121
 E r122
 E r123
 G r124
 L r125
 G r126
 G r127
 E r128
 E r129
 L r130
 G r131
 E r132
 G r133
 G r134
 L r135
 L r136
 E r137
 L r138
 E r139
 E r140
 G r141
 G r142
 L r143
 E r144
 L r145
 L r146
 E r147
 E r148
 G r149
COLUMNS
 x r5 7.0
 x r101 -2.0
 x r86 0.0
 x r105 0.0
 x r90 7.0
 x r91 -2.0
 x r72 -2.0
 x r76 0.0
 x r54 3.25
 x r133 0.001
 x r79 3.25
 x r138 -2.0
 x r130 -2.0
 x r69 0.001
 x r81 1.5
 x r66 0.001
 x r76 0.0
 x r40 1.5
 x r87 7.0
 x r63 0.001
 x r19 1.5
 x r71 1.5
 x r146 1.5
 x r147 7.0
 x r33 7.0
 x r2 3.25
 x r89 -2.0
 x r106 3.25
 x r145 3.25
 x r114 3.25
 x r44 0.0
 x r82 1.5
 x r45 7.0
 x r11 1.5
 x r94 7.0
 x r13 0.001
 x r140 3.25
 x r48 1.5
 x r119 -2.0
 x r123 3.25
 x r95 0.001
 x r9 -2.0
 x r139 0.001
 x r116 1.5
 x r96 1.5
 x r12 0.001
 x r89 3.25
 x r90 -2.0
 x r122 -2.0
 x r95 0.001
 x r26 7.0
 x r136 3.25
 x r16 0.0
 x r38 -2.0
 x r121 0.0
 x r6 0.0
 x r107 7.0
 x r148 0.001
 x r13 0.0
 x r46 3.25
 x r125 3.25
 x r20 -2.0
 x r8 0.001
 x r33 0.0
 x r18 0.001
 x r83 0.0
 x r71 0.001
 x r103 3.25
 x r108 0.0
 x r127 7.0
 x r144 1.5
 x r148 7.0
 x r68 7.0
 x r129 -2.0
 x r14 0.001
 x r42 -2.0
 x r52 3.25
 x r93 1.5
 x r48 1.5
 x r134 1.5
 x r25 1.5
 x r149 -2.0
 x r105 -2.0
 x r98 -2.0
 x r24 1.5
 x r25 3.25
 x r28 3.25
 x r143 0.001
 x r103 0.0
 x r0 1.5
 x r99 3.25
 x r50 3.25
 x r7 0.0
 x r141 7.0
 x r143 3.25
 x r102 0.0
 x r138 3.25
 x r10 3.25
 x r77 1.5
 x r147 -2.0
 x r112 -2.0
 x r117 0.001
 x r84 -2.0
 x r126 1.5
 x r60 0.0
 x r142 3.25
 x r80 3.25
 x r11 -2.0
 x r100 7.0
 x r57 0.0
 x r24 1.5
 x r36 0.0
 x r61 -2.0
 x r97 0.0
 x r118 0.001
 x r55 -2.0
 x r60 1.5
 x r43 3.25
 x r29 1.5
 x r27 0.0
 x r47 0.0
 x r15 0.0
 x r22 1.5
 x r70 0.001
 x r4 1.5
 x r30 0.001
 x r56 3.25
 x r62 3.25
 x r9 -2.0
 x r85 1.5
 x r39 0.001
 x r41 0.0
 x r1 0.0
 x r128 3.25
 x r114 0.001
 x r67 -2.0
 x r111 0.001
 x r64 -2.0
 x r17 1.5
 x r104 -2.0
 x r75 0.001
 x r86 1.5
 x r94 7.0
 x r131 -2.0
 x r88 0.001
 x r92 1.5
 x r21 7.0
 x obj 3.25
 x r59 0.001
 x r37 0.0
 x r137 0.001
 x r84 0.0
 x r110 -2.0
 x r23 3.25
 x r17 0.0
 x r3 0.0
 x r132 1.5
 x r78 0.001
 x r35 0.001
 x r115 3.25
 x r31 -2.0
 x r61 -2.0
 x r123 -2.0
 x r32 0.001
 x r58 -2.0
 x r104 1.5
 x r136 0.001
 x r73 7.0
 x r135 3.25
 x r120 0.001
 x r75 0.0
 x r74 1.5
 x r44 3.25
 x r20 1.5
 x r0 3.25
 x r49 0.001
 x r124 3.25
 x r34 -2.0
 x r109 1.5
 x r131 1.5
 x r18 7.0
 x r119 7.0
 x r51 7.0
 x r140 0.0
 x r65 1.5
 x obj 0.001
 x r77 1.5
 x obj -2.0
 x r72 3.25
 x r53 1.5
 x r29 3.25
 x r107 1.5
 x r113 0.001
 M1 'MARKER' 'INTORG'
 y r50 0.0
 y r140 -2.0
 y r58 3.25
 y r46 0.001
 y r66 0.001
 y r122 1.5
 y r97 -2.0
 y r31 3.25
 y r127 7.0
 y r65 3.25
 y r67 3.25
 y r23 1.5
 y r143 1.5
 y r136 7.0
 y r12 0.0
 y r90 -2.0
 y r12 0.001
 y r142 1.5
 y r139 3.25
 y r144 3.25
 y r16 0.0
 y r64 0.001
 y r88 1.5
 y r59 3.25
 y r149 0.001
 y r100 0.001
 y r96 0.0
 y r120 1.5
 y r15 1.5
 y r128 3.25
 y r38 0.0
 y r126 0.001
 y r121 -2.0
 y r56 0.0
 y r63 3.25
 y r75 -2.0
 y r9 0.0
 y r119 3.25
 y r80 0.001
 y r31 0.001
 y r99 1.5
 y r69 7.0
 y r130 1.5
 y r127 0.0
 y r146 3.25
 y r33 0.0
 y r25 1.5
 y r140 7.0
 y r35 3.25
 y r111 1.5
 y r81 -2.0
 y r30 3.25
 y r41 -2.0
 y r131 0.0
 y r36 0.001
 y r77 0.001
 y r40 1.5
 y r92 -2.0
 y r148 0.0
 y r94 1.5
 y r18 7.0
 y r114 3.25
 y r5 3.25
 y r133 -2.0
 y r118 7.0
 y r141 3.25
 y r135 -2.0
 y r136 1.5
 y r95 3.25
 y r82 0.0
 y r47 -2.0
 y r57 1.5
 y r26 1.5
 y r137 -2.0
 y obj 0.0
 y r32 1.5
 y r78 1.5
 y r134 0.0
 y r45 7.0
 y r97 3.25
 y r34 3.25
 y r148 7.0
 y r118 1.5
 y r84 1.5
 y r132 -2.0
 y r0 -2.0
 y r27 3.25
 y r29 1.5
 y r68 3.25
 y r43 7.0
 y r110 1.5
 y r98 0.0
 y r89 7.0
 y r66 0.001
 y r91 3.25
 y r87 7.0
 y r103 1.5
 M1 'MARKER' 'INTEND'
 y r86 1.5
 y r19 -2.0
 y r93 3.25
 y r100 7.0
 y r60 3.25
 y r24 7.0
 y r0 0.001
 y r14 0.001
 y r71 7.0
 y r117 0.001
 y r101 7.0
 y r109 1.5
 y r3 -2.0
 y r128 1.5
 y r53 3.25
 y r123 0.001
 y r107 3.25
 y r25 3.25
 y r51 3.25
 y r1 1.5
 y r62 -2.0
 y r60 0.0
 y r43 -2.0
 y r73 0.0
 y r72 -2.0
 y r27 3.25
 y r115 1.5
 y r147 0.0
 y r8 -2.0
 y r42 3.25
 y r30 3.25
 y r44 -2.0
 y r112 0.0
 y r74 -2.0
 y r6 3.25
 y r39 0.0
 y r108 0.0
 y r106 1.5
 y r41 0.001
 y r32 0.001
 y r125 1.5
 y r105 -2.0
 y r8 3.25
 y r48 1.5
 y r26 1.5
 y r124 0.001
 y r96 3.25
 y r107 1.5
 y r49 3.25
 y obj -2.0
 y r104 -2.0
 y r54 0.001
 y r103 1.5
 y r13 0.001
 y r79 3.25
 y r22 0.0
 y r105 7.0
 y r13 3.25
 y r83 -2.0
 y r85 3.25
 y r7 -2.0
 y r145 1.5
 y r17 0.0
 y r52 7.0
 y obj 1.5
 y r58 1.5
 y r113 -2.0
 y r21 3.25
 y r38 1.5
 y r90 0.0
 y r11 -2.0
 y r4 0.001
 y r71 0.0
 y r125 0.0
 y r45 -2.0
 y r2 -2.0
 y r129 0.0
 y r35 7.0
 y r20 0.001
 y r53 7.0
 y r116 1.5
 y r70 0.0
 y r55 1.5
 y r139 1.5
 y r102 0.001
 y r61 3.25
 y r138 0.0
 y r111 0.0
 y r46 0.0
 y r37 3.25
 y r10 7.0
 y r28 7.0
 y r76 3.25
 y r121 -2.0
 y r113 7.0
 y r123 -2.0
 z r19 0.001
 z r105 1.5
 z r100 7.0
 z r86 3.25
 z r32 3.25
 z obj 0.0
 z r53 1.5
 z r53 -2.0
 z r57 7.0
 z r82 1.5
 z r35 3.25
 z r31 1.5
 z r139 0.001
 z r101 -2.0
 z r88 -2.0
 z r143 -2.0
 z r94 7.0
 z r2 -2.0
 z r124 0.001
 z r120 3.25
 z r110 0.001
 z r114 1.5
 z r131 3.25
 z r63 3.25
 z r25 0.001
 z r140 -2.0
 z r60 0.001
 z r1 0.0
 z r12 1.5
 z r52 -2.0
 z r14 0.001
 z r101 -2.0
 z r128 7.0
 z r85 1.5
 z r145 0.001
 z r125 7.0
 z r112 0.0
 z r109 3.25
 z r149 7.0
 z r136 0.0
 z r121 -2.0
 z r115 0.001
 z r77 1.5
 z r90 0.0
 z r138 7.0
 z r80 7.0
 z r91 7.0
 z r84 3.25
 z r52 0.001
 z r68 0.001
 z r29 1.5
 z r39 0.0
 z r39 7.0
 z r78 1.5
 z r79 -2.0
 z r127 3.25
 z r50 1.5
 z r100 3.25
 z r34 1.5
 z r68 -2.0
 z r107 -2.0
 z r47 1.5
 z r17 -2.0
 z obj 3.25
 z r54 3.25
 z r148 3.25
 z obj 3.25
 z r71 0.0
 z r129 7.0
 z r64 1.5
 z r58 1.5
 z r6 0.001
 z r56 7.0
 z r4 7.0
 z r65 3.25
 z r13 7.0
 z r37 7.0
 z r40 7.0
 z r49 0.001
 z r116 7.0
 z r99 -2.0
 z r97 1.5
 z r55 0.001
 z r69 3.25
 z r7 7.0
 z r43 -2.0
 z r76 1.5
 z r134 0.001
 z r128 0.0
 z r16 -2.0
 z r84 1.5
 z r59 7.0
 z r74 0.0
 z r135 1.5
 z r144 -2.0
 z r30 -2.0
 z r21 3.25
 z r119 1.5
 z r98 1.5
 z r66 3.25
 z r102 1.5
 z r115 0.0
 z r72 -2.0
 z r4 0.001
 z r118 3.25
 z r88 3.25
 z r30 -2.0
 z r93 1.5
 z r80 3.25
 z r24 0.001
 z r73 -2.0
 z r125 -2.0
 z r42 -2.0
 z r38 3.25
 z r50 3.25
 z r92 -2.0
 z r33 3.25
 z r20 1.5
 z r0 0.0
 z r3 3.25
 z r43 7.0
 z r113 -2.0
 z r104 0.001
 z r133 0.001
 z r111 0.001
 z r35 0.0
 z r46 1.5
 z r123 -2.0
 z r114 3.25
 z r83 3.25
 z r108 0.001
 z r137 0.001
 z r117 7.0
 z r61 0.001
 z r32 0.001
 z r117 3.25
 z r41 0.001
 z r142 -2.0
 z r10 -2.0
 z r103 0.0
 z r21 7.0
 z r15 0.0
 z r18 0.001
 z r67 0.001
 z r9 1.5
 z r90 -2.0
 z r95 7.0
 z r36 -2.0
 z r89 0.0
 z r5 0.001
 z r55 1.5
 z r22 0.001
 z r108 7.0
 z r75 3.25
 z r34 0.0
 z r132 -2.0
 z r20 0.001
 z r81 0.0
 z r148 -2.0
 z r63 1.5
 z r87 7.0
 z r23 -2.0
 z r70 0.001
 z r8 3.25
 z r132 1.5
 z r22 0.001
 z r27 1.5
 z r146 0.001
 z r126 7.0
 z r147 3.25
 z r67 0.0
 z r18 0.001
 z r27 1.5
 z r51 3.25
 z r106 -2.0
 z r44 0.0
 z r45 1.5
 z r72 1.5
 z r62 0.001
 z r64 7.0
 z r44 0.0
 z r112 3.25
 z r71 0.0
 z r130 7.0
 z r141 3.25
 z r70 0.0
 z r48 7.0
 z r96 7.0
 z r28 3.25
 z r122 -2.0
 z r38 3.25
 z r11 3.25
 z r26 7.0
 w r46 -2.0
 w r110 7.0
 w r101 0.001
 w r44 0.0
 w r106 0.0
 w r33 7.0
 w r37 0.001
 w r81 -2.0
 w r95 0.0
 w r31 3.25
 w r119 7.0
 w r37 7.0
 w r131 0.001
 w r29 0.001
 w r46 -2.0
 w r146 3.25
 w r102 0.001
 w r123 0.001
 w r40 0.001
 w r69 -2.0
 w r78 7.0
 w r51 7.0
 w r15 7.0
 w r1 0.001
 w r21 1.5
 w obj 0.0
 w r130 -2.0
 w r36 3.25
 w r88 3.25
 w r120 3.25
 w r138 7.0
 w r142 0.001
 w r74 -2.0
 w r26 3.25
 w r73 3.25
 w r10 3.25
 w r112 3.25
 w r117 0.001
 w r87 7.0
 w r84 -2.0
 w r78 -2.0
 w r57 0.0
 w r122 0.001
 w r99 0.001
 w r103 -2.0
 w r54 -2.0
 w r33 1.5
 w r63 1.5
 w r58 3.25
 w r104 0.001
 w r56 3.25
 w r44 7.0
 w r5 0.001
 w r82 1.5
 w r9 0.001
 w r58 -2.0
 w r149 1.5
 w r147 3.25
 w r144 0.001
 w r75 7.0
 w r93 0.001
 w r60 0.001
 w r11 -2.0
 w r97 1.5
 w r70 1.5
 w r104 1.5
 w r142 3.25
 w r20 3.25
 w r100 -2.0
 w r65 1.5
 w r114 1.5
 w r6 0.0
 w r71 3.25
 w r69 7.0
 w r62 0.001
 w r129 7.0
 w r50 0.001
 w r43 0.0
 w r136 0.001
 w r47 -2.0
 w r18 -2.0
 w r48 3.25
 w r115 7.0
 w r53 3.25
 w r117 0.001
 w r52 7.0
 w r2 0.0
 w r76 1.5
 w r89 7.0
 w r139 0.0
 w r141 1.5
 w r4 3.25
 w r125 1.5
 w r38 -2.0
 w r55 -2.0
 w r132 7.0
 w r134 0.0
 w r105 0.0
 w r0 0.0
 w r148 -2.0
 w r102 0.001
 w r45 -2.0
 w r124 3.25
 w r24 0.001
 w r2 1.5
 w r55 7.0
 w r19 3.25
 w r66 3.25
 w r133 -2.0
 w r65 0.0
 w r145 3.25
 w r28 0.0
 w r19 -2.0
 w r118 0.0
 w r85 3.25
 w r80 1.5
 w r29 0.0
 w r67 1.5
 w r91 -2.0
 w r67 -2.0
 w r87 0.0
 w r61 0.0
 w r91 -2.0
 w r116 0.0
 w r14 -2.0
 w r28 0.0
 w r15 1.5
 w r64 0.001
 w r22 7.0
 w r113 3.25
 w r7 0.0
 w r90 1.5
 w r127 7.0
 w r77 7.0
 w r49 0.001
 w r109 0.0
 w r111 1.5
 w r27 -2.0
 w r92 0.001
 w r66 1.5
 w r32 1.5
 w r107 3.25
 w r86 0.0
 w r94 -2.0
 w r24 7.0
 w r49 -2.0
 w r23 0.0
 w r35 0.0
 w r126 -2.0
 w r36 1.5
 w r12 3.25
 w r112 0.001
 w r30 3.25
 w r115 3.25
 w r84 1.5
 w r68 1.5
 w r59 0.0
 w r105 7.0
 w r8 0.001
 w r3 3.25
 w r79 0.001
 w r98 1.5
 w r140 1.5
 w r108 1.5
 w r42 0.0
 w r13 0.0
 w r96 7.0
 w r121 3.25
 w obj -2.0
 w r25 0.001
 w r83 0.001
 w r143 -2.0
 w r39 0.0
 w r34 7.0
 w obj 0.0
 w r72 0.0
 w r17 1.5
 w r16 3.25
 w r94 3.25
 w r135 7.0
 w r99 1.5
 w r48 3.25
 w r141 3.25
 w r51 0.0
 w r41 0.001
 w r137 3.25
 w r71 1.5
 w r128 0.0
 w r79 7.0
 w r80 1.5
 w r4 -2.0
 w r31 3.25
 w r116 1.5
RHS
 rhs r0 0
 rhs r1 1
 rhs r2 2
 rhs r3 3
 rhs r4 4
 rhs r5 5
 rhs r6 6
 rhs r7 0
 rhs r8 1
 rhs r9 2
 rhs r10 3
 rhs r11 4
 rhs r12 5
 rhs r13 6
 rhs r14 0
 rhs r15 1
 rhs r16 2
 rhs r17 3
 rhs r18 4
 rhs r19 5
 rhs r20 6
 rhs r21 0
 rhs r22 1
 rhs r23 2
 rhs r24 3
 rhs r25 4
 rhs r26 5
 rhs r27 6
 rhs r28 0
 rhs r29 1
 rhs r30 2
 rhs r31 3
 rhs r32 4
 rhs r33 5
 rhs r34 6
 rhs r35 0
 rhs r36 1
 rhs r37 2
 rhs r38 3
 rhs r39 4
 rhs r40 5
 rhs r41 6
 rhs r42 0
 rhs r43 1
 rhs r44 2
 rhs r45 3
 rhs r46 4
 rhs r47 5
 rhs r48 6
 rhs r49 0
 rhs r50 1
 rhs r51 2
 rhs r52 3
 rhs r53 4
 rhs r54 5
 rhs r55 6
 rhs r56 0
 rhs r57 1
 rhs r58 2
 rhs r59 3
 rhs r60 4
 rhs r61 5
 rhs r62 6
 rhs r63 0
 rhs r64 1
 rhs r65 2
 rhs r66 3
 rhs r67 4
 rhs r68 5
 rhs r69 6
 rhs r70 0
 rhs r71 1
 rhs r72 2
 rhs r73 3
 rhs r74 4
 rhs r75 5
 rhs r76 6
 rhs r77 0
 rhs r78 1
 rhs r79 2
 rhs r80 3
 rhs r81 4
 rhs r82 5
 rhs r83 6
 rhs r84 0
 rhs r85 1
 rhs r86 2
 rhs r87 3
 rhs r88 4
 rhs r89 5
 rhs r90 6
 rhs r91 0
 rhs r92 1
 rhs r93 2
 rhs r94 3
 rhs r95 4
 rhs r96 5
 rhs r97 6
 rhs r98 0
 rhs r99 1
 rhs r100 2
 rhs r101 3
 rhs r102 4
 rhs r103 5
 rhs r104 6
 rhs r105 0
 rhs r106 1
 rhs r107 2
 rhs r108 3
 rhs r109 4
 rhs r110 5
 rhs r111 6
 rhs r112 0
 rhs r113 1
 rhs r114 2
 rhs r115 3
 rhs r116 4
 rhs r117 5
 rhs r118 6
 rhs r119 0
 rhs r120 1
 rhs r121 2
 rhs r122 3
 rhs r123 4
 rhs r124 5
 rhs r125 6
 rhs r126 0
 rhs r127 1
 rhs r128 2
 rhs r129 3
 rhs r130 4
 rhs r131 5
 rhs r132 6
 rhs r133 0
 rhs r134 1
 rhs r135 2
 rhs r136 3
 rhs r137 4
 rhs r138 5
 rhs r139 6
 rhs r140 0
 rhs r141 1
 rhs r142 2
 rhs r143 3
 rhs r144 4
 rhs r145 5
 rhs r146 6
 rhs r147 0
 rhs r148 1
 rhs r149 2
BOUNDS
 UP b y 9
 UP b w 3
ENDATA
