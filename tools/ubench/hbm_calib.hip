// hbm_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access
// shapes the step kernels use (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  Every kernel touches each byte of its buffer exactly once; buffers are > 512 MiB
// (past the 256 MiB Infinity Cache), so reported / actual is the counter's scale for that shape.
//   calib_read4 / calib_write4       4 B per lane, lane-contiguous (the [E][N] arrays: rewards, distances, actions)
//   calib_read16 / calib_write16     16 B per lane, lane-contiguous (observation rows: float4 stores; the guide's own shape)
//   calib_read_pose / calib_write_pose   the lane-group kernel's pose access: a group of 8 lanes owns one env, lanes 0..4
//       touch x, y, theta of agents 0..4 of a 60-byte [3][5] block -- three 4-byte accesses per lane, 5 of 8 lanes active,
//       ENVS_PER_WAVE env slots of a wave in use (4 at the headline's 4096 envs), one wave per workgroup, XCD-aware chunk map
//   calib_write1                     1 B per lane (done / violation flags)
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/ubench/libhbm_calib.so tools/ubench/hbm_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void calib_read4(const float *src, size_t n, float *sink) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        acc += src[i];
    if (acc == 123456.789f) sink[0] = acc;  // never true for the zero-filled buffer: keeps the loads alive
}
__global__ void calib_read16(const float4 *src, size_t n, float *sink) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) sink[0] = acc;
}
__global__ void calib_write4(float *dst, size_t n) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[i] = 1.0f;
}
__global__ void calib_write16(float4 *dst, size_t n) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}
__global__ void calib_write1(uint8_t *dst, size_t n) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[i] = 1;
}

// the step kernels' block -> env-chunk map (csrc/device_common.h xcd_chunk)
__device__ __forceinline__ int xcd_chunk(int G) {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, q = G >> 3, r = G & 7;
    return xcd * q + (xcd < r ? xcd : r) + j;
}
template <bool WRITE>
__global__ __launch_bounds__(64) void calib_pose(float *poses, int E, int epw, float *sink) {
    const int lane = threadIdx.x, ag = lane & 7, g = lane >> 3;
    const int e = xcd_chunk(gridDim.x) * epw + g;
    if (g >= epw || e >= E || ag >= 5) return;
    float *X = poses + static_cast<size_t>(e) * 15;
    if (WRITE) {
        X[ag] = 1.0f;
        X[5 + ag] = 2.0f;
        X[10 + ag] = 3.0f;
    } else {
        const float acc = X[ag] + X[5 + ag] + X[10 + ag];
        if (acc == 123456.789f) sink[0] = acc;
    }
}

extern "C" {
// mode: 0 read4, 1 read16, 2 write4, 3 write16, 4 read_pose, 5 write_pose, 6 write1.  `bytes` = size of buf; for the pose
// modes E = bytes / 60 envs and `epw` env slots per wave (1, 2, 4 or 8).  Returns the bytes the kernel touches.
long long calib_run(int mode, void *buf, long long bytes, int epw, void *sink, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int blocks = 256 * 16, threads = 256;
    switch (mode) {
        case 0: hipLaunchKernelGGL(calib_read4, dim3(blocks), dim3(threads), 0, s, static_cast<const float *>(buf), static_cast<size_t>(bytes / 4), static_cast<float *>(sink)); return bytes / 4 * 4;
        case 1: hipLaunchKernelGGL(calib_read16, dim3(blocks), dim3(threads), 0, s, static_cast<const float4 *>(buf), static_cast<size_t>(bytes / 16), static_cast<float *>(sink)); return bytes / 16 * 16;
        case 2: hipLaunchKernelGGL(calib_write4, dim3(blocks), dim3(threads), 0, s, static_cast<float *>(buf), static_cast<size_t>(bytes / 4)); return bytes / 4 * 4;
        case 3: hipLaunchKernelGGL(calib_write16, dim3(blocks), dim3(threads), 0, s, static_cast<float4 *>(buf), static_cast<size_t>(bytes / 16)); return bytes / 16 * 16;
        case 6: hipLaunchKernelGGL(calib_write1, dim3(blocks), dim3(threads), 0, s, static_cast<uint8_t *>(buf), static_cast<size_t>(bytes)); return bytes;
        case 4: case 5: {
            const int E = static_cast<int>(bytes / 60);
            const int grid = (E + epw - 1) / epw;
            if (mode == 4) hipLaunchKernelGGL(calib_pose<false>, dim3(grid), dim3(64), 0, s, static_cast<float *>(buf), E, epw, static_cast<float *>(sink));
            else hipLaunchKernelGGL(calib_pose<true>, dim3(grid), dim3(64), 0, s, static_cast<float *>(buf), E, epw, static_cast<float *>(sink));
            return static_cast<long long>(E) * 60;
        }
        default: return -1;
    }
}
}
