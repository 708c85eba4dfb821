// Error string + version entry points.
#include "dmt_common.h"

static thread_local char g_err[512] = "";

void dmt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dmt_last_error(void) { return g_err; }

// Deterministic mode: every reduction that normally combines partial sums with fp32 atomics (order = scheduling) takes a fixed-order
// form instead -- slower, bit-reproducible from run to run.
static int g_deterministic = 0;
int dmt_deterministic(void) { return g_deterministic; }
extern "C" int dmt_set_deterministic(int32_t on) { g_deterministic = on ? 1 : 0; return DMT_OK; }
extern "C" int dmt_get_deterministic(void) { return g_deterministic; }
extern "C" int dmt_version(void) { return 1; }
extern "C" const char* dmt_build_arch(void) { return "gfx950"; }

// sizeof() of every ABI struct, so bindings in other languages can verify their layout (tests/test_abi.py).
extern "C" int dmt_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(dmt_gather_feature);
    case 1: return (int)sizeof(dmt_gather_desc);
    case 2: return (int)sizeof(dmt_embgrad_desc);
    case 3: return (int)sizeof(dmt_gemm_desc);
    case 4: return (int)sizeof(dmt_attn_desc);
    case 5: return (int)sizeof(dmt_attn_bwd_desc);
    case 6: return (int)sizeof(dmt_table_map);
    case 7: return (int)sizeof(dmt_cast_job);
    case 8: return (int)sizeof(dmt_chain_desc);
    case 9: return (int)sizeof(dmt_wgrad_desc);
    case 10: return (int)sizeof(dmt_mhsa_desc);
    case 11: return (int)sizeof(dmt_mmoe_desc);
    case 12: return (int)sizeof(dmt_heads_desc);
    case 13: return (int)sizeof(dmt_q1mem_desc);
    default: return -1;
  }
}
