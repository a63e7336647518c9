// placeholder until the MFMA attention lands (entry points return UNSUPPORTED)
#include "common.h"
extern "C" int fcmf_attn_mfma_fwd(const void*, const void*, const void*, const float*, void*, float*, int, int, int, int,
                                  int64_t, int64_t, int64_t, float, float, uint64_t, void*) { return FCMF_ERR_UNSUPPORTED; }
extern "C" int fcmf_attn_mfma_bwd(const void*, const void*, const void*, const float*, const void*, const void*, const float*,
                                  void*, void*, void*, int, int, int, int, int64_t, int64_t, int64_t, float, float, uint64_t,
                                  void*) { return FCMF_ERR_UNSUPPORTED; }
