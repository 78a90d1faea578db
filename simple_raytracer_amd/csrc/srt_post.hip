// srt_post.hip -- the post-process filters of the finished picture: Sorting.BlurArray, ContrastBlurArray and Despeckle
// (MiscFunctions.cs:124-191, :193-335, :337-416; Game1.cs:226-232 runs Despeckle blurPasses times).
//
// The reference filters work IN PLACE in row-major order (`Color[] originalArray = array` is an alias, not a copy): the
// new value of pixel (x, y) is computed from the NEW values of (x-1, y) and of row y-1 and from the OLD values of
// (x+1, y) and of row y+1.  That is a recurrence, not a stencil; what it leaves free is the skewed order: all pixels
// with the same x + 2y can be computed together, because everything (x, y) reads was finished at x + 2y - 1 at the
// latest and everything that reads its old value comes later.
//
// post_band_kernel cuts the interior rows into bands of 128; every band is one workgroup, every row one thread.  At step
// t the thread of local row r handles column 1 + t - 2r.  It keeps the old values it needs (its own row and the row
// below, from the untouched input picture) in registers as a sliding window, its left neighbour's new value in a
// register, and takes the three new values of the row above from a 4-deep ring in LDS that the thread above wrote one
// to three steps earlier; one barrier per step.  The bands of a picture run at the same time, each behind the one above
// it: the first row of a band reads the row above from the output picture, and the thread of a band's last row
// publishes every 16 columns (release store of a counter in global memory) how far that row is; thread 0 of the band
// below waits (acquire loads) until the columns it needs are there.  Nothing waits for a band below, so the bands form a
// pipeline that cannot lock up as long as all of them are resident, which the launcher sees to (at most 512 workgroups
// per launch); a poll count bounds the wait all the same and raises an alarm the host turns into an error.  The output
// picture starts as a copy of the input, and a pixel is written only when its value changes.
//
// It is a latency-bound kernel by construction (about W + 2H dependent steps): 8 bytes read and at most 4 written per
// pixel, far from any roofline.  The arithmetic is the reference's, float for float: bytes to float,
// the sum of the eight neighbours left to right in the order the source adds them, Vector3 / 8 as a multiplication by
// 0.125, `new Color(v / 255)` as a true division followed by (int)(c * 255) with clamping.
#include <hip/hip_runtime.h>

#include "../../include/srt.h"
#include "srt_launch.h"
#include "srt_post.h"

namespace srt {

namespace {

constexpr int kBandRows = 128;                        // rows per band = threads of a workgroup
constexpr int kPostPublish = 16;                      // a band reports the progress of its last row every so many columns
constexpr int kMaxBands = 512;                        // workgroups of one launch: all resident at once, so none waits for one that cannot start
constexpr unsigned long long kPostMaxPolls = 1ull << 26;   // watchdog of the wait for the band above

template <int FILTER>
__global__ void __launch_bounds__(kBandRows) post_band_kernel(const uint32_t* __restrict__ in, uint32_t* out, int width, int height, int first_row,
                                                               float weight, float param, uint32_t* progress)
{
    __shared__ uint32_t ring[kBandRows][4];   // new values of the last four columns of every row of the band
    const int band = (int)blockIdx.x, row0 = first_row + band * kBandRows;
    const int rows = height - 1 - row0 < kBandRows ? height - 1 - row0 : kBandRows;
    const int r = (int)threadIdx.x, y = row0 + r;
    const bool mine = r < rows;
    const uint32_t* own = in + (size_t)(mine ? y : row0) * (size_t)width;
    const uint32_t* below = own + width;
    const uint32_t* above_in = own - width;
    const uint32_t* above_out = out + (size_t)(row0 - 1) * (size_t)width;   // the last row of the band above (or row 0)
    // the window of old values slides along with the thread's column
    uint32_t cur = 0u, right = 0u, d_left = 0u, d_mid = 0u, d_right = 0u, left_new = 0u;
    if (mine) { cur = own[1]; right = own[2]; d_left = below[0]; d_mid = below[1]; d_right = below[2]; left_new = own[0]; }
    uint32_t known = band == 0 ? 0x7FFFFFFFu : 0u;   // thread 0: columns the band above is known to have finished in its last row
    const int steps = (width - 2) + 2 * (rows - 1);
    for (int t = 0; t < steps; t++) {
        const int x = 1 + t - 2 * r;
        if (mine && x >= 1 && x <= width - 2) {
            Window w;
            w.left = left_new; w.right = right;
            w.down = d_mid; w.left_down = d_left; w.right_down = d_right;
            if (r == 0) {
                // needs columns x - 1 .. x + 1 of the row above in their final state
                const uint32_t need = (uint32_t)(x + 1 < width - 2 ? x + 1 : width - 2);
                if (known < need) {
                    unsigned long long polls = 0ull;
                    do {
                        known = __hip_atomic_load(&progress[band - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        // Giving up is final: the alarm is raised once, this band stops waiting for good, and its own
                        // progress word reports every column as done, so the bands below leave as well instead of
                        // each spinning through the poll budget for every column.
                        if (++polls > kPostMaxPolls || __hip_atomic_load(&progress[kMaxBands], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                            __hip_atomic_store(&progress[kMaxBands], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&progress[band], 0x7FFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            known = 0x7FFFFFFFu;
                        }
                    } while (known < need);
                }
                w.left_up = above_out[x - 1]; w.up = above_out[x]; w.right_up = above_out[x + 1];
            } else {
                // columns 0 and width - 1 are never filtered: their "new" value is the input's
                w.left_up = x - 1 == 0 ? above_in[0] : ring[r - 1][(x - 1) & 3];
                w.up = ring[r - 1][x & 3];
                w.right_up = x + 1 == width - 1 ? above_in[width - 1] : ring[r - 1][(x + 1) & 3];
            }
            const uint32_t now = filtered<FILTER>(cur, w, weight, param);
            if (now != cur) out[(size_t)y * (size_t)width + (size_t)x] = now;
            ring[r][x & 3] = now;
            left_new = now;
            // the band below reads this band's last row: tell it every kPostPublish columns how far the row is
            if (r == rows - 1 && ((x & (kPostPublish - 1)) == 0 || x == width - 2))
                __hip_atomic_store(&progress[band], (uint32_t)x, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            // slide: the loads are for the next step and have the barrier to arrive in
            cur = right;
            d_left = d_mid; d_mid = d_right;
            if (x + 2 <= width - 1) { right = own[x + 2]; d_right = below[x + 2]; }
        }
        __syncthreads();
    }
}

}  // namespace

size_t post_progress_bytes() { return (size_t)(kMaxBands + 1) * sizeof(uint32_t); }

hipError_t launch_post_filter(int filter, const uint32_t* d_in, uint32_t* d_out, int width, int height, float weight, float param,
                              uint32_t* d_progress, hipStream_t stream)
{
    hipError_t e = hipMemcpyAsync(d_out, d_in, (size_t)width * (size_t)height * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    if (width < 3 || height < 3) return hipSuccess;   // no interior pixel
    weight = weight < 0.0f ? 0.0f : (weight > 1.0f ? 1.0f : weight);   // Math.Clamp(weight, 0, 1)
    for (int row0 = 1; row0 <= height - 2; row0 += kMaxBands * kBandRows) {
        const int rows = height - 1 - row0;
        int bands = (rows + kBandRows - 1) / kBandRows;
        if (bands > kMaxBands) bands = kMaxBands;
        // progress[b]: last finished column of band b's last row; progress[kMaxBands]: the watchdog's alarm, which the
        // caller clears once and reads at the end
        e = hipMemsetAsync(d_progress, 0, (size_t)bands * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        if (filter == SRT_POST_BLUR) hipLaunchKernelGGL((post_band_kernel<SRT_POST_BLUR>), dim3(bands), dim3(kBandRows), 0, stream, d_in, d_out, width, height, row0, weight, param, d_progress);
        else if (filter == SRT_POST_CONTRAST_BLUR) hipLaunchKernelGGL((post_band_kernel<SRT_POST_CONTRAST_BLUR>), dim3(bands), dim3(kBandRows), 0, stream, d_in, d_out, width, height, row0, weight, param, d_progress);
        else hipLaunchKernelGGL((post_band_kernel<SRT_POST_DESPECKLE>), dim3(bands), dim3(kBandRows), 0, stream, d_in, d_out, width, height, row0, weight, param, d_progress);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace srt
