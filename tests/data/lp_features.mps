* exercises every card type the reader handles
NAME  FEATURES
OBJSENSE
    MIN
ROWS
 N  OBJ
 N  RIMOBJ
 E  EQP
 E  EQN
 L  LE1
 G  GE1
 G  GE2
COLUMNS
    A  OBJ  1.5  EQP  1.0
    A  LE1  2.0
    A  LE1  0.5
    A  RIMOBJ  9.0
    B  EQN  -1.0  GE1  4.0
    MARKER  'MARKER'  'INTORG'
    C  GE2  1.0  OBJ  -2.0
    MARKER  'MARKER'  'INTEND'
    D  EQP  3.0  GE2  -1.0
    E  LE1  1.0
    F  GE1  1.0
    G  EQN  2.0
    H  GE2  5.0
RHS
    R  OBJ  -7.0  EQP  4.0
    R  EQN  -2.0  LE1  10.0
    R  GE1  1.0
    OTHER  GE2  99.0
RANGES
    RG  EQP  2.0  EQN  -3.0
    RG  LE1  4.0  GE1  -6.0
BOUNDS
 FR BD  A
 MI BD  B
 UP BD  D  -1.0
 UP BD  E  5.0
 LO BD  F  2.0
 FX BD  G  3.5
 BV BD  H
 UP OTHERSET  A  1.0
ENDATA
