// tps_kernels.hip -- one group of scan kernels of libtopsicle_hip.so (see tps_kernels.h); compiled with -DTPS_KGROUP=<n>.
#ifndef TPS_KGROUP
#error "compile with -DTPS_KGROUP=<1 .. TPS_KGROUPS - 1>"
#endif
#include "tps_kernels.h"
