// libqsparse_hip.so -- the channels_last staged-mean kernels for f16 inputs (qs_mean_cl_host.h; entry points: api_mean_cl.hip)
#include "qs_mean_cl_host.h"

QS_MEAN_CL_DTYPE_UNIT(f16, QS_F16)
