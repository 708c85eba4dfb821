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


// Launch-route trace (diagnostic, off by default): while on, every successful launch counts under its route label -- the string its
// DMT_CHECK_LAUNCH names, e.g. "dmt_attn_fwd(mfma, coalesced)" -- so a test can assert WHICH kernel variant an entry point took.
#include <string.h>
#include <mutex>
int g_dmt_route_trace = 0;
namespace {
struct RouteSlot { char label[64]; long long count; };
RouteSlot g_routes[128];
int g_n_routes = 0;
std::mutex g_route_mu;
}  // namespace
void dmt_route_note(const char* what) {
  std::lock_guard<std::mutex> lk(g_route_mu);
  for (int i = 0; i < g_n_routes; ++i)
    if (strncmp(g_routes[i].label, what, sizeof(g_routes[i].label) - 1) == 0) { ++g_routes[i].count; return; }
  if (g_n_routes < 128) {
    strncpy(g_routes[g_n_routes].label, what, sizeof(g_routes[0].label) - 1);
    g_routes[g_n_routes].label[sizeof(g_routes[0].label) - 1] = 0;
    g_routes[g_n_routes++].count = 1;
  }
}
extern "C" int dmt_route_trace(int32_t on) {
  std::lock_guard<std::mutex> lk(g_route_mu);
  if (on) g_n_routes = 0;
  g_dmt_route_trace = on ? 1 : 0;
  return DMT_OK;
}
extern "C" int64_t dmt_route_count(const char* label) {
  if (!label) return -1;
  std::lock_guard<std::mutex> lk(g_route_mu);
  for (int i = 0; i < g_n_routes; ++i)
    if (strcmp(g_routes[i].label, label) == 0) return g_routes[i].count;
  return 0;
}
extern "C" int dmt_route_dump(char* buf, int32_t cap) {
  if (!buf || cap <= 0) return DMT_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_route_mu);
  int off = 0;
  buf[0] = 0;
  for (int i = 0; i < g_n_routes; ++i) {
    const int n = snprintf(buf + off, (size_t)(cap - off), "%s=%lld\n", g_routes[i].label, g_routes[i].count);
    if (n < 0 || n >= cap - off) break;
    off += n;
  }
  return DMT_OK;
}
extern "C" int dmt_version(void) { return DMT_ABI_VERSION; }
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
    case 14: return (int)sizeof(dmt_mhsa_bwd_desc);
    default: return -1;
  }
}
