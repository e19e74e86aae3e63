// t2d_host.h -- host-side helpers shared by the translation units of libt2d_hip.so's C ABI (t2d_api.hip: pool lifetime, launches,
// transfers; t2d_geometry_host.hip: the reference's polygons -> what the kernels consume).  Not part of the ABI.
#pragma once
#include <string>
#include <vector>

#include "t2d_pool.h"

namespace t2d {
namespace host {

// error text of a failing call: on the pool, or (pool == nullptr) of the thread's last failing t2d_create / host-only call
int fail(t2d_pool* p, int code, const std::string& msg);
const std::string& create_error();

#define T2D_HIP(p, call)                                                                                  \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return ::t2d::host::fail(p, T2D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// replace a device buffer by a copy of n host elements (n == 0: free it)
template <class T>
int dev_replace(t2d_pool* p, T** dst, const T* src, size_t n) {
    if (*dst) {
        T2D_HIP(p, hipFree(*dst));
        *dst = nullptr;
    }
    if (n == 0) return T2D_OK;
    T2D_HIP(p, hipMalloc((void**)dst, n * sizeof(T)));
    T2D_HIP(p, hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return T2D_OK;
}


// ---- geometry preparation (t2d_geometry_host.hip) -----------------------------------------------------------------------------
double area2(const std::vector<double>& P);
double orient_h(const double* p, const double* q, const double* r);
int log2_pad(int A);
int envs_per_workgroup(t2d_pool* p, int log2A);
int prepare_polys(t2d_pool* p, const int32_t* env_off, const int32_t* vert_off, const float* xy, t2d_pool::HostGeo& out);
void build_lane_boundary(int E, t2d_pool::HostGeo& g);
void build_safe_rects(int E, t2d_pool::HostGeo& g);
void fill_layout(t2d::GeoLayout& gl, int epb, const int mp[2], const int mv[2], int mb = 0);
int rebuild_geo(t2d_pool* p);

}  // namespace host
}  // namespace t2d
