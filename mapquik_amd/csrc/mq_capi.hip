// mq_capi.hip -- kernels + the extern "C" boundary declared in include/mapquik_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see mapquik_amd/build.py).  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mq_device.hpp"
#include "mq_seed.hpp"
#include "mq_fastx.hpp"

using namespace mq;

// The library is ONE translation unit (the kernels inline the device headers; the entry points share static state), kept in parts:
#include "mq_map_kernels.hpp"    // map_kernel and its split-pipeline twins
#include "mq_build_kernels.hpp"  // index build, on-disk form, lookup
#include "mq_host_state.hpp"     // mq_index, mq_ctx, geometry, scratch
#include "mq_capi_index.hpp"     // mq_index_new .. mq_index_finalize
#include "mq_capi_index_io.hpp"  // save / load / clone
#include "mq_capi_map.hpp"       // contexts, map entry points, FASTA chunks, PAF, host memory
#include "mq_capi_diag.hpp"      // include/mapquik_hip_diag.h
