// Timing-only experiment hooks for elvis_amd/csrc/conv.hip (NOT product code; most of these switches make the
// kernel produce wrong results on purpose).  Selected with -DELVIS_EXP_* by tools/build_variant.py:
//   NOSTAGE  no staging traffic in the K loop        NOW  no weight staging      NOH  no halo staging
//   NOBARRIER  row-step barriers removed             NOSTORE  epilogue stores skipped
//   SETPRIO  s_setprio(1) around the MFMA clusters   STAMP  s_memtime phase stamps written over the stats slot
#pragma once
#define ELVIS_CONV_HOOKS 1
#ifdef ELVIS_EXP_NOSTAGE
#define ELVIS_STAGE(x)
#else
#define ELVIS_STAGE(x) x
#endif
#ifdef ELVIS_EXP_NOW
#define ELVIS_STAGE_W(x)
#else
#define ELVIS_STAGE_W(x) ELVIS_STAGE(x)
#endif
#ifdef ELVIS_EXP_NOH
#define ELVIS_STAGE_H(x)
#else
#define ELVIS_STAGE_H(x) ELVIS_STAGE(x)
#endif
#ifdef ELVIS_EXP_SETPRIO
#define ELVIS_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define ELVIS_SETPRIO(x)
#endif
#ifdef ELVIS_EXP_NOBARRIER
#define ELVIS_BARRIER()
#else
#define ELVIS_BARRIER() __syncthreads()
#endif
#ifdef ELVIS_EXP_NOSTORE   /* stores skipped at run time (the condition is never false), values kept alive */
#define ELVIS_HOOK_SKIP_STORE(tv) if (p.cout_pitch < 0x7ffffff0) { asm volatile("" :: "v"(tv[0]), "v"(tv[1]), "v"(tv[2]), "v"(tv[3])); } else
#else
#define ELVIS_HOOK_SKIP_STORE(tv)
#endif
#ifdef ELVIS_EXP_STAMP     /* per-workgroup phase cycle counts, written over the tile's statistics slot */
#define ELVIS_HOOK_STAMP_BEGIN const unsigned long long stamp0 = __builtin_amdgcn_s_memtime(); unsigned long long stamp1 = 0;
#define ELVIS_HOOK_STAMP_LOOP stamp1 = __builtin_amdgcn_s_memtime();
#define ELVIS_HOOK_STAMP_EPILOGUE const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();
#define ELVIS_HOOK_STAMP_END                                                                               \
    if (p.stats) {                                                                                         \
        const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        const unsigned long long stamp4 = __builtin_amdgcn_s_memtime();                                    \
        __syncthreads();                                                                                   \
        if (tid == 0) {                                                                                    \
            long long tile = ((long long)nimg * p.tiles_y + ty) * p.tiles_x + tx;                          \
            float* dst = p.stats + (tile * p.cout + co0) * 2;                                              \
            dst[0] = (float)(stamp1 - stamp0); dst[1] = (float)(stamp2 - stamp1);                          \
            dst[2] = (float)(stamp3 - stamp2); dst[3] = (float)(stamp4 - stamp3);                          \
            dst[4] = (float)(stamp0 & 0xffffff); dst[5] = (float)__builtin_amdgcn_s_memrealtime();         \
        }                                                                                                  \
    }
#else
#define ELVIS_HOOK_STAMP_BEGIN
#define ELVIS_HOOK_STAMP_LOOP
#define ELVIS_HOOK_STAMP_EPILOGUE
#define ELVIS_HOOK_STAMP_END
#endif
