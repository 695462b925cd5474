// nufft_hip_build_info(): what this binary is. One line of `key=value` pairs separated by ';':
//   abi=<NUFFT_HIP_ABI_VERSION>;source=<first 16 hex digits of the SHA-256 over the library's sources, in the
//   Makefile's order>;arch=gfx950;experiment=<none | the experiment macros in force with their values>
// The digest depends on the tree, not on the commit: a rebuild of unchanged sources is byte-identical, and a test
// recomputes it from csrc/ to tell a stale library from a current one. Compiled with the same -D flags as the
// other translation units (Makefile EXTRA, tools/variant_build.sh).
#include "nufft_experiment.h"
#include "nufft_hip.h"

#define NUFFT_STR2(x) #x
#define NUFFT_STR(x) NUFFT_STR2(x)

#ifndef NUFFT_SOURCE_DIGEST
#define NUFFT_SOURCE_DIGEST unknown
#endif

extern "C" const char* nufft_hip_build_info(void) {
  return "abi=" NUFFT_STR(NUFFT_HIP_ABI_VERSION) ";source=" NUFFT_STR(NUFFT_SOURCE_DIGEST) ";arch=gfx950;experiment="
#ifndef NUFFT_EXPERIMENT_MACROS_IN_FORCE
#ifdef NUFFT_EXPERIMENT_BUILD
         "declared"   // -DNUFFT_EXPERIMENT_BUILD without any of its macros: still not a product build
#else
         "none"
#endif
#else
         "yes"
#ifdef NUFFT_GROUP_EXP
         ",NUFFT_GROUP_EXP=" NUFFT_STR(NUFFT_GROUP_EXP)
#endif
#ifdef NUFFT_DENSE_EXP
         ",NUFFT_DENSE_EXP=" NUFFT_STR(NUFFT_DENSE_EXP)
#endif
#ifdef NUFFT_INTERP_EXP
         ",NUFFT_INTERP_EXP=" NUFFT_STR(NUFFT_INTERP_EXP)
#endif
#ifdef NUFFT_GROUP_NW
         ",NUFFT_GROUP_NW=" NUFFT_STR(NUFFT_GROUP_NW)
#endif
#ifdef NUFFT_GROUP_STAGE
         ",NUFFT_GROUP_STAGE=" NUFFT_STR(NUFFT_GROUP_STAGE)
#endif
#ifdef NUFFT_DENSE_NW
         ",NUFFT_DENSE_NW=" NUFFT_STR(NUFFT_DENSE_NW)
#endif
#ifdef NUFFT_PATCH_NW
         ",NUFFT_PATCH_NW=" NUFFT_STR(NUFFT_PATCH_NW)
#endif
#ifdef NUFFT_PATCH_MINW
         ",NUFFT_PATCH_MINW=" NUFFT_STR(NUFFT_PATCH_MINW)
#endif
#ifdef NUFFT_STACK_ROWS
         ",NUFFT_STACK_ROWS=" NUFFT_STR(NUFFT_STACK_ROWS)
#endif
#ifdef NUFFT_BOUND_THREADS
         ",NUFFT_BOUND_THREADS=" NUFFT_STR(NUFFT_BOUND_THREADS)
#endif
#ifdef NUFFT_FX_BOUND_LIMIT
         ",NUFFT_FX_BOUND_LIMIT=" NUFFT_STR(NUFFT_FX_BOUND_LIMIT)
#endif
#ifdef NUFFT_HIP_NO_PRELOAD
         ",NUFFT_HIP_NO_PRELOAD"
#endif
#ifdef NUFFT_HIP_PHASE_LOG
         ",NUFFT_HIP_PHASE_LOG"
#endif
#ifdef NUFFT_MIX_SHAPE_ENV
         ",NUFFT_MIX_SHAPE_ENV"
#endif
#endif
      ;
}
