// Detector-crop extraction for the batched multi-object estimator:
// AePoseEstimator.extract_square_patch(black_borders=True) followed by
// cv2.resize(..., INTER_LINEAR)  (/root/reference/auto_pose/m3_interface/ae_pose_estimator.py:106-131,
// 157-162), for ALL detections of an image in one launch, uint8 in / uint8 out.
//
// Per detection (x, y, w, h, size = int(max(h, w) * pad_factor)): a size x size black canvas with
// the box pixels pasted at its centre is resized to OH x OW with OpenCV's uint8 bilinear
// arithmetic (11-bit fixed-point coefficients; horizontal taps clamped with zeroed fraction,
// vertical rows clipped; dst = ((b0*(h0>>4))>>16 + (b1*(h1>>4))>>16 + 2) >> 2).  The canvas is
// never materialised: every output pixel gathers its four taps straight from the image.
// HBM-bound byte work: one thread per output pixel, channels innermost (coalesced stores).
#pragma once

namespace aae {

struct CropResizeArgs {
    const unsigned char* img;   // [H][W][C]
    const int* boxes;           // [D][5] = x, y, w, h, size
    unsigned char* out;         // [D][OH][OW][C]
    int H, W, C, D, OH, OW;
};

__device__ __forceinline__ void cv_linear_tap(int d, int src, int dst, bool horizontal, int& s, int& c0, int& c1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    c0 = (int)__builtin_rintf((1.f - f) * 2048.f);
    c1 = (int)__builtin_rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void crop_resize_bilinear_u8_kernel(const CropResizeArgs p) {
    const int d = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p.OH * p.OW) return;
    const int dy = pix / p.OW, dx = pix - dy * p.OW;
    const int bx = p.boxes[d * 5 + 0], by = p.boxes[d * 5 + 1], bw = p.boxes[d * 5 + 2], bh = p.boxes[d * 5 + 3];
    const int size = p.boxes[d * 5 + 4];
    unsigned char* o = p.out + (((long long)d * p.OH + dy) * p.OW + dx) * p.C;
    if (size < 1) {
        for (int ch = 0; ch < p.C; ++ch) o[ch] = 0;
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    cv_linear_tap(dx, size, p.OW, true, sx, a0, a1);
    cv_linear_tap(dy, size, p.OH, false, sy, b0, b1);
    const int x0 = sx, x1 = min(sx + 1, size - 1);
    const int y0 = min(max(sy, 0), size - 1), y1 = min(max(sy + 1, 0), size - 1);
    const int oy = (size - bh) >> 1, ox = (size - bw) >> 1;    // floor division, as python's //
    // canvas (r, c) -> image pixel, or black
    auto tap = [&](int r, int c, int ch) -> int {
        const int rr = r - oy, cc = c - ox;
        if (rr < 0 || rr >= bh || cc < 0 || cc >= bw) return 0;
        const int iy = by + rr, ix = bx + cc;
        if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) return 0;
        return (int)p.img[((long long)iy * p.W + ix) * p.C + ch];
    };
    for (int ch = 0; ch < p.C; ++ch) {
        const int h0 = tap(y0, x0, ch) * a0 + tap(y0, x1, ch) * a1;
        const int h1 = tap(y1, x0, ch) * a0 + tap(y1, x1, ch) * a1;
        int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        o[ch] = (unsigned char)v;
    }
}

}  // namespace aae
