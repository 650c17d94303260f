def UniTuple(t, n):
    return ("UniTuple", t, n)


int64 = "int64"
int32 = "int32"
float64 = "float64"
