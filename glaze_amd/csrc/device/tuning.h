// Every compile-time switch of the render kernels, in one place.  tools/build_variant.sh NAME "-DGLZ_X=..." builds a library with one
// of them changed (variants/libglaze_hip_NAME.so, selected at run time with GLAZE_HIP_LIB); the product is built with the defaults.
// What each value was measured against is in EXPERIMENTS.md; switches whose experiment ended in "no" are gone from the sources
// (rounds 1-3 carried 37 of them), their result is in EXPERIMENTS.md as well.
//
//   switch                 default  what
//   ---------------------  -------  ----------------------------------------------------------------------------------------------
//   GLZ_TRACE_WAVES           6     waves per SIMD k_trace is compiled for (80 VGPRs)
//   GLZ_TRACE_SPLIT           1     k_trace: with more groups than waves, waves take groups of ONE kind (closest-hit or shadow); _NUM / _DEN (9 / 10): the shadow waves' share relative to the shadow groups' share
//   GLZ_TRACE_PREFETCH        0     k_trace: the next node's loads issued as soon as the node is known (as k_path does; wants 16 more registers)
//   GLZ_TRACE_TL_WAVES        4     ... the two-level tracer (128 VGPRs)
//   GLZ_SHADE_WAVES           4     ... k_shade (128 VGPRs)
//   GLZ_PATH_WAVES            4     ... k_path (128 VGPRs)
//   GLZ_TRACE8_WAVES          4     ... k_trace8, the 8-wide walk of a small tile share (128 VGPRs)
//   GLZ_TRACE8_PREFETCH       1     k_trace8: the next node's loads issued as soon as the node is known
//   GLZ_TRACE8_STACK         28     k_trace8: stack levels kept in LDS (1 KB a level and block)
//   GLZ_REFILL               10     idle lanes at which a wave takes new rays (16 until the camera rays left the refill: round 5)
//   GLZ_TL_REFILL            16     the same for the two-level tracer (8: 0.834 against 0.827 ms, round 3)
//   GLZ_LEAF_QUORUM          24     lanes waiting on a leaf at which the inner-node phase ends
//   GLZ_LEAF_QUORUM_ANY      30     ... in a pass of shadow rays only (18 / 24 / 30 / 36 / 42: 0.444 / 0.432 / 0.429 / 0.431 / 0.434 ms per k_trace); GLZ_REFILL_ANY: = GLZ_REFILL (6 / 16: no difference)
//   GLZ_TL_LEAF_QUORUM       32     the same for the two-level tracer (a leaf visit there is an instance entry: dearer)
//   GLZ_ALPHA_QUORUM         12     lanes waiting with a candidate on non-opaque geometry at which the alpha phase runs
//   GLZ_PATH_PREFETCH         1     k_path: the next node's loads issued as soon as the node is known (trace_wave<PREFETCH>)
//   GLZ_NODE48               off    EXPERIMENT: 48-byte nodes (types.h BvhNode48; tools/build_variant_full.sh: scene.cpp needs it too)
//   GLZ_NO_SHARE / _ANY / _CLOSEST, GLZ_NO_LDS_TOP   off   debugging: the tail's work sharing off (both passes / the shadow pass / the closest-hit pass), every node from memory
//   GLZ_WAVE_TIMES           off    instrumentation: per-wave time stamps of k_trace            (tools/gpu_wave_times.py)
//   GLZ_PATH_TIMES           off    instrumentation: per-wave tracing / shading time of k_path  (tools/gpu_path_phases.py)
//   GLZ_SECTION_TIMES        off    instrumentation: clocks per part of trace_wave's round; = 2: the node visit in pieces too (tools/gpu_sections.py)
#pragma once
#ifndef GLZ_TRACE_WAVES
#define GLZ_TRACE_WAVES 6
#endif
#ifndef GLZ_TRACE_SPLIT
#define GLZ_TRACE_SPLIT 1
#endif
#ifndef GLZ_TRACE_SPLIT_NUM
#define GLZ_TRACE_SPLIT_NUM 9
#endif
#ifndef GLZ_TRACE_SPLIT_DEN
#define GLZ_TRACE_SPLIT_DEN 10
#endif
#ifndef GLZ_TRACE_PREFETCH
#define GLZ_TRACE_PREFETCH 0
#endif
#ifndef GLZ_TRACE_TL_WAVES
#define GLZ_TRACE_TL_WAVES 4
#endif
#ifndef GLZ_SHADE_WAVES
#define GLZ_SHADE_WAVES 4
#endif
#ifndef GLZ_PATH_WAVES
#define GLZ_PATH_WAVES 4
#endif
#ifndef GLZ_TRACE8_WAVES
#define GLZ_TRACE8_WAVES 4
#endif
#ifndef GLZ_TRACE8_STACK
#define GLZ_TRACE8_STACK 28
#endif
#ifndef GLZ_TRACE8_PREFETCH
#define GLZ_TRACE8_PREFETCH 1
#endif
#ifndef GLZ_REFILL
#define GLZ_REFILL 10
#endif
#ifndef GLZ_REFILL_ANY
#define GLZ_REFILL_ANY GLZ_REFILL
#endif
#ifndef GLZ_LEAF_QUORUM_ANY
#define GLZ_LEAF_QUORUM_ANY 30
#endif
#ifndef GLZ_TL_REFILL
#define GLZ_TL_REFILL 16
#endif
#ifndef GLZ_LEAF_QUORUM
#define GLZ_LEAF_QUORUM 24
#endif
#ifndef GLZ_ALPHA_QUORUM
#define GLZ_ALPHA_QUORUM 12
#endif
#ifndef GLZ_TL_LEAF_QUORUM
#define GLZ_TL_LEAF_QUORUM 32
#endif
#ifndef GLZ_PATH_PREFETCH
#define GLZ_PATH_PREFETCH 1
#endif
