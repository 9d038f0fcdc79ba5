// cssm_series_abi.h -- constants of the persistent series kernel shared by its device code (cssm_series.hip.h) and the
// host driver (cssm_pf.hip), which must not instantiate the kernels itself.
#pragma once
#define CSSM_SER_GROUP 32            /* blocks per arrival group of the barrier */
#define CSSM_SER_MAXBLOCKS 1024      /* 32 groups of 32: lanes 0..31 of one wave read the groups, lanes 32..63 the own group */
#define CSSM_SER_LW_CAP 2048         /* log-weights a block can keep in LDS between the phases */
#define CSSM_SER_SPIN_LIMIT (1u << 22)
#define CSSM_SER_ABORT (1ull << 63)
// timestamps of block 0 (profiling): 5 per observation -- start, end of phase P, end of the exchange, end of phase O,
// end of the closing barrier -- in ticks of the constant 100 MHz counter (s_memrealtime)
#define CSSM_SER_TS_PER_STEP 5
