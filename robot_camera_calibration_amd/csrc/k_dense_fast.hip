// k_dense_fast.hip -- placeholder translation unit; the row-marching variant of the dense pass
// is added here (see k_dense.hip for the weak defaults that report "not supported").
#include "rcc_internal.h"
