// gaq_inst.hip -- the step / rollout kernel instantiations of part GAQ_PART (0 ... 7) of gaq_kernels.hpp's lists: compiled eight
// times side by side (Makefile), linked with gaq.o into libgaq.so.
#include "gaq_kernels.hpp"

#ifndef GAQ_PART
#error "compile with -DGAQ_PART=0 ... 7"
#endif
#define GAQ_CAT2(a, b) a##b
#define GAQ_CAT(a, b) GAQ_CAT2(a, b)
#define GAQ_X(FEAT) template __global__ GAQ_STEP_SIG(FEAT)
GAQ_CAT(GAQ_STEP_PART, GAQ_PART)(GAQ_X)
#undef GAQ_X
#define GAQ_X(FEAT) template __global__ GAQ_ROLL_SIG(FEAT)
GAQ_CAT(GAQ_ROLL_PART, GAQ_PART)(GAQ_X)
#undef GAQ_X
