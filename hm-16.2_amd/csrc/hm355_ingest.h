// hm355 picture ingest and output: TVideoIOYuv::read (TVideoIOYuv.cpp:633-695: readPlane :227 + scalePlane :70) and
// TVideoIOYuv::write (:706-792: scalePlane + writePlane :362) for planar 4:2:0 files, on the device: the raw frame bytes go to HBM as they
// are on disk and one launch turns them into the slot's original planes (bit-depth scaling, right / bottom padding by repetition) -- or the
// slot's reconstruction into file bytes (conformance-window crop, rounding and clipping when the file is shallower).  Pure HBM streaming:
// one lane per 8 consecutive samples of a row (16 B of plane data, one vector load and one vector store when the file row is aligned), lanes
// along the rows of a plane.
#pragma once

struct IngestParams {                 // one picture of the batch
  const uint8_t *src; uint8_t *dst;   // raw frame in / raw frame out (device)
  int32_t fileW, fileH, fileBitDepth; // ingest: size and depth of the file frame; output: cropped size and depth of the file frame
  int32_t fromOrg;                    // output: 0 = the reconstruction, 1 = the original planes
};

// scalePlane :70-88 (CLIP_TO_709_RANGE off): shift > 0 multiplies, shift < 0 divides with rounding and clips to [0, maxval]
__device__ __forceinline__ int hm_yuv_scale(int v, int shift, int maxval)
{
  if (shift > 0) return (int)(int16_t)(v << shift);
  if (shift < 0) { const int r = (v + (1 << (-shift - 1))) >> (-shift); return r < 0 ? 0 : (r > maxval ? maxval : r); }
  return v;
}

// grid (ceil(groups of the luma plane / 256), 1, 3 * n): blockIdx.z = picture * 3 + component; a group = 8 consecutive samples of a row
#define HM_INGEST_BLOCK 256
extern "C" __global__ void __launch_bounds__(HM_INGEST_BLOCK) hm355_ingest_kernel(const Params *P, const IngestParams *ips)
{
  const int f = (int)blockIdx.z / 3, c = (int)blockIdx.z % 3, cs = c ? 1 : 0;
  const IngestParams ip = ips[f];
  const int w = P->width >> cs, h = P->height >> cs, fw = ip.fileW >> cs, fh = ip.fileH >> cs;
  const int gpr = (w + 7) >> 3, idx = (int)blockIdx.x * HM_INGEST_BLOCK + (int)threadIdx.x;
  if (idx >= gpr * h) return;
  const int y = idx / gpr, x0 = (idx - y * gpr) * 8;
  const int is16 = ip.fileBitDepth > 8, bps = is16 ? 2 : 1, shift = P->bitDepth - ip.fileBitDepth, maxval = (1 << P->bitDepth) - 1;
  // plane offsets inside the file frame: Y, then Cb, then Cr (4:2:0)
  const size_t lumaBytes = (size_t)ip.fileW * ip.fileH * bps, planeOff = c == 0 ? 0 : lumaBytes + (size_t)(c - 1) * (lumaBytes >> 2);
  const int sy = y < fh ? y : fh - 1;                                   // bottom padding repeats the last row
  const uint8_t *row = ip.src + planeOff + (size_t)sy * fw * bps;
  Pel *dst = P->frames[f].org[c] + (size_t)y * P->stride[c] + x0;
  alignas(16) Pel v[8];
  const uint8_t *p8 = row + (size_t)x0 * bps;
  if (x0 + 8 <= fw && is16 && (((uintptr_t)p8) & 15) == 0) {            // whole group inside the file row, 16-byte aligned: one load
    alignas(16) uint16_t t[8]; *(uint4 *)t = *(const uint4 *)p8;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (Pel)hm_yuv_scale((int)(int16_t)t[i], shift, maxval);
  } else if (x0 + 8 <= fw && !is16 && (((uintptr_t)p8) & 7) == 0) {
    alignas(8) uint8_t t[8]; *(uint2 *)t = *(const uint2 *)p8;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (Pel)hm_yuv_scale((int)t[i], shift, maxval);
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int x = x0 + i, sx = x < fw ? x : fw - 1;                   // right padding repeats the last sample
      const int s = is16 ? (int)(int16_t)((uint32_t)row[2 * sx] | ((uint32_t)row[2 * sx + 1] << 8)) : (int)row[sx];
      v[i] = (Pel)hm_yuv_scale(s, shift, maxval);
    }
  }
  if (x0 + 8 <= w) *(uint4 *)dst = *(const uint4 *)v;                   // rows of the slot's planes are 128-byte aligned
  else for (int i = 0; x0 + i < w; i++) dst[i] = v[i];
}

// the slot's reconstruction (or original planes) as a file frame: same grid over the cropped picture
extern "C" __global__ void __launch_bounds__(HM_INGEST_BLOCK) hm355_output_kernel(const Params *P, const IngestParams *ips)
{
  const int f = (int)blockIdx.z / 3, c = (int)blockIdx.z % 3, cs = c ? 1 : 0;
  const IngestParams ip = ips[f];
  const int fw = ip.fileW >> cs, fh = ip.fileH >> cs;
  const int gpr = (fw + 7) >> 3, idx = (int)blockIdx.x * HM_INGEST_BLOCK + (int)threadIdx.x;
  if (idx >= gpr * fh) return;
  const int y = idx / gpr, x0 = (idx - y * gpr) * 8;
  const int is16 = ip.fileBitDepth > 8, bps = is16 ? 2 : 1, shift = ip.fileBitDepth - P->bitDepth, maxval = (1 << ip.fileBitDepth) - 1;
  const size_t lumaBytes = (size_t)ip.fileW * ip.fileH * bps, planeOff = c == 0 ? 0 : lumaBytes + (size_t)(c - 1) * (lumaBytes >> 2);
  const Pel *src = (ip.fromOrg ? P->frames[f].org[c] : P->frames[f].rec[c]) + (size_t)y * P->stride[c] + x0;
  uint8_t *row = ip.dst + planeOff + (size_t)y * fw * bps;
  alignas(16) Pel v[8];
  *(uint4 *)v = *(const uint4 *)src;                                    // the slot's rows are padded to whole CTUs: always readable
  uint8_t *p8 = row + (size_t)x0 * bps;
  if (x0 + 8 <= fw && is16 && (((uintptr_t)p8) & 15) == 0) {
    alignas(16) uint16_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = (uint16_t)hm_yuv_scale((int)v[i], shift, maxval);
    *(uint4 *)p8 = *(const uint4 *)t;
  } else if (x0 + 8 <= fw && !is16 && (((uintptr_t)p8) & 7) == 0) {
    alignas(8) uint8_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = (uint8_t)hm_yuv_scale((int)v[i], shift, maxval);
    *(uint2 *)p8 = *(const uint2 *)t;
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int x = x0 + i;
      if (x < fw) {
        const int s = hm_yuv_scale((int)v[i], shift, maxval);
        if (is16) { row[2 * x] = (uint8_t)(s & 0xff); row[2 * x + 1] = (uint8_t)((s >> 8) & 0xff); } else row[x] = (uint8_t)s;
      }
    }
  }
}
