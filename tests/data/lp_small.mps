* 2x2 LP with the known answer x=(2.8,3.6), obj=-26.4 (same LP as the reference's example; file written for this repo)
NAME          SMALL2X2
ROWS
 N  COST
 L  LIM1
 L  LIM2
COLUMNS
    X1  LIM1  1.0  LIM2  3.0
    X1  COST  -3.0
    X2  LIM1  2.0  LIM2  1.0
    X2  COST  -5.0
RHS
    B  LIM1  10.0  LIM2  12.0
BOUNDS
 LO BND  X1  0.0
 LO BND  X2  0.0
ENDATA
