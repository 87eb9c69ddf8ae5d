// mhx_kernels.hip -- the ahead-of-time compiled gfx950 kernels: device code in
// mhx_kernels.hpp (shared with the run-time compiled user-expression kernels), spec table and
// launchers in mhx_launch.inc.
#include "mhx_kernels.hpp"

#include "mhx_launch.inc"
