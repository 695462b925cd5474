// Experiment builds (tools/variant_build.sh, tools/*_experiment.sh).
//
// The macros listed here re-shape a product kernel (waves per workgroup, staging depth, thread counts, a fixed bound
// limit) or REMOVE pieces of its main loop ("wrong results, timing only": the knock-out builds behind
// EXPERIMENTS.md section 11). None of them belongs in a shipped library, so defining any of them on the compiler command
// line requires -DNUFFT_EXPERIMENT_BUILD as well; without it the translation unit does not compile. The macros in
// force are reported by nufft_hip_build_info() (nufft_build_info.cpp), and a CPU test refuses a shipped library that
// reports any. Included first by every translation unit (through nufft_hip_internal.h / nufft_device.h).
#ifndef NUFFT_EXPERIMENT_H_
#define NUFFT_EXPERIMENT_H_

#if defined(NUFFT_GROUP_EXP) || defined(NUFFT_DENSE_EXP) || defined(NUFFT_INTERP_EXP) || defined(NUFFT_GROUP_NW) || \
    defined(NUFFT_GROUP_STAGE) || defined(NUFFT_DENSE_NW) || defined(NUFFT_PATCH_NW) || defined(NUFFT_PATCH_MINW) ||  \
    defined(NUFFT_STACK_ROWS) || defined(NUFFT_BOUND_THREADS) || defined(NUFFT_FX_BOUND_LIMIT) ||                     \
    defined(NUFFT_HIP_NO_PRELOAD) || defined(NUFFT_HIP_PHASE_LOG) || defined(NUFFT_MIX_SHAPE_ENV)
#define NUFFT_EXPERIMENT_MACROS_IN_FORCE 1
#ifndef NUFFT_EXPERIMENT_BUILD
#error "an experiment macro (NUFFT_*_EXP, NUFFT_*_NW, NUFFT_HIP_PHASE_LOG, ...) is defined: such builds change kernel shapes or drop work and must say so with -DNUFFT_EXPERIMENT_BUILD (see nufft_experiment.h)"
#endif
#endif

#endif  // NUFFT_EXPERIMENT_H_
