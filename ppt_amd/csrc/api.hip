// api.hip -- error strings / ABI version of libppt_hip.so.
#include "ppt_common.h"

extern "C" const char *ppt_strerror(int code)
{
    switch (code) {
    case PPT_OK: return "ok";
    case PPT_EINVAL: return "invalid argument (shape, alignment or NULL pointer)";
    case PPT_ELAUNCH: return "kernel launch failed (hipGetLastError)";
    case PPT_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown ppt error";
    }
}

extern "C" int ppt_abi_version(void) { return 7; }   // 2: + ppt_bn_finalize_ws, ppt_rows_stats_f32, ppt_bn_rows_bwd_*; 3: PPT_F16 operands (dtype fields / arguments), ppt_cross_entropy_rows ignore + scale, ppt_adamw_step grad_scale, ppt_bn_rows_bwd_apply half_dtype; 4: see include/ppt_hip.h

static thread_local int g_wave_priority = 0;
extern "C" void ppt_set_wave_priority(int prio) { g_wave_priority = prio > 0 ? 1 : 0; }
extern "C" int ppt_get_wave_priority(void) { return g_wave_priority; }

// -1: the environment's PPT_GEMM256 (default on); 0 / 1: ppt_gemm's automatic use of the 256-row macro-tile core off / on
static thread_local int g_gemm256 = -1;
extern "C" void ppt_set_gemm256(int mode) { g_gemm256 = mode < 0 ? -1 : (mode ? 1 : 0); }
extern "C" int ppt_get_gemm256(void) { return g_gemm256; }

static thread_local int g_persistent_percent = 100;
extern "C" void ppt_set_persistent_occupancy(int percent) { g_persistent_percent = percent < 10 ? 10 : (percent > 100 ? 100 : percent); }
extern "C" int ppt_get_persistent_occupancy(void) { return g_persistent_percent; }
