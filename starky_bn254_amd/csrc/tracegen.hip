// placeholder, replaced below
#include "host_common.hpp"
using namespace sbn;
extern "C" int sbn_generate_trace_g1_exp(const uint32_t*, size_t, uint64_t*, uint64_t*) { return fail(SBN_ERR_UNSUPPORTED, "not built yet"); }
extern "C" int sbn_generate_trace_g1_op(const uint32_t*, size_t, uint64_t*) { return fail(SBN_ERR_UNSUPPORTED, "not built yet"); }
