// Translation unit of the software-pipelined d = 40 attention kernel (attention3_kernel.h); built with -fno-honor-nans
// (Makefile), which is why it is not part of attention.hip.
#include "attention3_kernel.h"
