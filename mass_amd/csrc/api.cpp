// api.cpp — version, error string and device-property cache of libmassfuse.so
#include "common.h"

namespace mf {

char *error_buffer()
{
    static thread_local char buf[512] = "";
    return buf;
}

const DeviceInfo &device_info()
{
    // one entry per device ordinal; filled on first use from the calling thread
    static DeviceInfo info[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (info[dev].cus == 0) {
        hipDeviceProp_t prop;
        DeviceInfo d = {256, 160 * 1024};   // MI355X defaults if the query fails
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            if (prop.multiProcessorCount > 0) d.cus = prop.multiProcessorCount;
            if (prop.maxSharedMemoryPerMultiProcessor > 0) d.lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
        }
        info[dev] = d;
    }
    return info[dev];
}

}  // namespace mf

extern "C" {

int mf_version(void) { return MF_ABI_VERSION; }

const char *mf_last_error(void) { return mf::error_buffer(); }

int mf_struct_sizes(size_t *grid_bytes, size_t *frames_bytes)
{
    if (grid_bytes) *grid_bytes = sizeof(mf_grid);
    if (frames_bytes) *frames_bytes = sizeof(mf_frames);
    return MF_OK;
}

}
