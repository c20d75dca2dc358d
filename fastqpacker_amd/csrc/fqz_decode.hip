// placeholder until the decode kernels land
#include "fqz_ctx.h"
int fqz_dec_launch(fqz_ctx *, const uint8_t *, size_t, uint8_t, int, uint8_t *, size_t, hipStream_t) { return FQZ_E_ARG; }
int fqz_dec_finish(fqz_ctx *, fqz_batch_result *) { return FQZ_E_ARG; }
int fqz_dec_entropy_only(fqz_ctx *, const uint8_t *, size_t, uint8_t *, size_t, size_t *, hipStream_t) { return FQZ_E_ARG; }
