#include "filler_kernels.h"
