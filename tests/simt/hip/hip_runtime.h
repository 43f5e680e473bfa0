// <hip/hip_runtime.h> of the host-side SIMT emulation (tests/simt/simt_hip.h)
#pragma once
#include "../simt_hip.h"
