// libqsparse_hip.so -- the NCHW staged-mean kernels for bf16 inputs (qs_mean_host.h; entry points: api_mean.hip)
#include "qs_mean_host.h"

QS_MEAN_DTYPE_UNIT(bf16, QS_BF16)
