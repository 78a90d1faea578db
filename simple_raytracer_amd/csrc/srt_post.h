// srt_post.h -- one pixel of the post-process filters Sorting.BlurArray / ContrastBlurArray / Despeckle
// (MiscFunctions.cs:124-191, :193-335, :337-416), float for float as the reference computes it.  Shared by the GPU
// kernel (srt_post.hip) and the test-suite's CPU harness, which runs it over the picture in the reference's row-major
// in-place order.
#ifndef SRT_POST_H
#define SRT_POST_H

#include "../../include/srt.h"
#include "srt_core.h"

namespace srt {

SRT_HD f3 bytes_of(uint32_t c) { return mk3((float)(c & 255u), (float)((c >> 8) & 255u), (float)((c >> 16) & 255u)); }
SRT_HD uint32_t color_of(f3 v) { return pack_color(v.x / 255.0f, v.y / 255.0f, v.z / 255.0f); }
SRT_HD float mixdown(f3 c) { return c.x * 0.3f + c.y * 0.6f + c.z * 0.11f; }   // MiscFunctions.cs:231, :364
SRT_HD f3 lerp_to(f3 original, f3 other, float weight)   // (originalColor * (1 - weight)) + (other * weight)
{
    return add3(scale3(original, 1.0f - weight), scale3(other, weight));
}

struct Window { uint32_t left, right, up, down, left_up, right_up, left_down, right_down; };

// tempColorLeft + tempColorRight + tempColorUp + tempColorDown + tempColorLeftUp + tempColorLeftDown + tempColorRightUp
// + tempColorRightDown, then / 8 (MiscFunctions.cs:181, :402)
SRT_HD f3 mean_of_eight(const Window& w)
{
    f3 s = add3(bytes_of(w.left), bytes_of(w.right));
    s = add3(s, bytes_of(w.up));
    s = add3(s, bytes_of(w.down));
    s = add3(s, bytes_of(w.left_up));
    s = add3(s, bytes_of(w.left_down));
    s = add3(s, bytes_of(w.right_up));
    s = add3(s, bytes_of(w.right_down));
    return scale3(s, 0.125f);
}

template <int FILTER>
SRT_HD uint32_t filtered(uint32_t current, const Window& w, float weight, float param)
{
    const f3 original = bytes_of(current);
    if (FILTER == SRT_POST_BLUR) {
        return color_of(lerp_to(original, mean_of_eight(w), weight));
    } else if (FILTER == SRT_POST_DESPECKLE) {
        const float gray = mixdown(original);
        if (!(gray <= 0.0f || gray > param)) return current;                          // :399
        const f3 blur = lerp_to(original, mean_of_eight(w), weight);                   // :401-402
        return color_of(lerp_to(original, blur, weight));                              // :404-407
    } else {
        // ContrastBlurArray: the neighbours whose gray value differs from the pixel's by more than `contrast`, in the
        // order the source collects them; starting from the first, every one that differs from the first is mixed in
        // half and half (:290-322)
        const float gray = mixdown(original);
        const uint32_t order[8] = {w.left, w.right, w.up, w.down, w.left_up, w.left_down, w.right_up, w.right_down};
        bool any = false;
        f3 first = original, blur = original;
        for (int i = 0; i < 8; i++) {
            const f3 c = bytes_of(order[i]);
            const float g = mixdown(c);
            const float diff = gray >= g ? gray - g : g - gray;
            if (!(fabsf(diff) > param)) continue;
            if (!any) { any = true; first = c; blur = c; }
            else if (c.x != first.x || c.y != first.y || c.z != first.z) blur = add3(scale3(blur, 0.5f), scale3(c, 0.5f));
        }
        return color_of(lerp_to(original, blur, weight));
    }
}

}  // namespace srt
#endif  // SRT_POST_H
