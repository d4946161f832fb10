/*
 * vstab.h -- C ABI of the MI355X-native stabilisation / fisheye-undistort hot path.
 *
 * Drop-in boundary for the reference's FrameSource -> FrameSourceWarp stage
 * (reference paths are relative to /root/reference/opencv/).  Plain pointers and sizes only;
 * no C++, OpenCV, HIP or torch types cross this interface.  `stream` arguments are a
 * hipStream_t passed as void* (NULL = the default stream); every device pointer must belong
 * to the device that is current on the calling thread.  Stateless entry points enqueue work
 * on `stream` and return without synchronising unless they say otherwise.
 *
 * Error model: the reference throws `int` (EOF == -1 for end of stream,
 * FrameSourceWarp.cpp:466; other values for failures, :181,:303).  Nothing is thrown across
 * this ABI; the same information travels as vstab_status return codes and
 * vstab_last_error().  include/vstab_frame_source.hpp re-throws them as `int` so C++ callers
 * see the reference's behaviour.
 */
#ifndef VSTAB_H_
#define VSTAB_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSTAB_API __attribute__((visibility("default")))

typedef enum vstab_status {
    VSTAB_OK = 0,
    VSTAB_EOF = -1,          /* == EOF thrown at FrameSourceWarp.cpp:466 */
    VSTAB_ERR_INVALID = -2,  /* bad argument / unsupported geometry */
    VSTAB_ERR_DEVICE = -3,   /* HIP runtime error (details in vstab_last_error) */
    VSTAB_ERR_NOMEM = -4,
    VSTAB_ERR_SOURCE = -5,   /* upstream callback failed with a code other than EOF */
    VSTAB_ERR_UNSUPPORTED = -6 /* a valid request this entry point does not serve (the text says which one does) */
} vstab_status;

/* CameraPreset, FrameSourceWarp.hpp:14-21 (same order, same values) */
typedef enum vstab_camera_preset {
    VSTAB_GOPRO_H4B_WIDE43_PUBLISHED = 0,
    VSTAB_GOPRO_H4B_WIDE43_MEASURED = 1,
    VSTAB_GOPRO_H4B_WIDE43_MEASURED_STABILISATION = 2,
    VSTAB_GOPRO_H4B_WIDE169_PUBLISHED = 3,
    VSTAB_GOPRO_H4B_WIDE169_MEASURED = 4,
    VSTAB_GOPRO_H4B_WIDE169_MEASURED_STABILISATION = 5
} vstab_camera_preset;

/* Thread-local text of the last failure of any call on this thread. */
VSTAB_API const char *vstab_last_error(void);
/* "vstab <version> gfx950" -- also proves the HIP code object is embedded. */
VSTAB_API const char *vstab_version(void);
/* Number of HIP devices visible, or a negative vstab_status. */
VSTAB_API int vstab_device_count(void);
/* sizeof() of the ABI structs as this library was compiled, for bindings to check their mirrors against:
 * 0 vstab_frame, 1 vstab_source, 2 vstab_config, 3 vstab_frame_log, 4 vstab_profile; -1 for any other index. */
VSTAB_API int vstab_struct_size(int which);
/* Layout version of the structs above.  It changes whenever one of them gains, loses or re-orders a member; a caller built against
 * another version of this header must not share a vstab_config / vstab_frame with this library.  vstab_config carries it
 * (vstab_config.abi_version, written by vstab_config_default) and vstab_create refuses any other value -- so a vstab_config MUST be
 * initialised with vstab_config_default() and then modified, never zero-filled or assembled by hand.  vstab_frame is always
 * allocated and zeroed by the library before a source callback fills it in.  Version 4: abi_version, vstab_frame.dmabuf_modifier;
 * map_precision defaults to VSTAB_MAP_PRECISION_OPENCL.  Version 5: vstab_config.read_ahead, two more counters in vstab_profile.
 * The value is "VSB" + the version in the low byte: no field an older layout had at this offset (a preset 0..5) can hold it, so a
 * struct filled in against an older header is refused whatever its contents -- and with it every later call that would write a
 * larger vstab_profile into that caller's smaller one, since no handle is ever created for it. */
#define VSTAB_ABI_VERSION 0x56534205
VSTAB_API int vstab_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Cameras (host, fp64).  Replaces get_preset_camera / get_output_camera,
 * FrameSourceWarp.cpp:27-86 and :88-165, including their integer quirks.
 * Matrices are row-major 3x3 doubles.
 * ------------------------------------------------------------------------------------------ */
VSTAB_API vstab_status vstab_get_preset_camera(int preset, int width, int height, double K[9]);
VSTAB_API vstab_status vstab_get_output_camera(const double K_in[9], int width, int height,
                                               double scale, int crop_borders, double zoom,
                                               double K_out[9], int *out_width, int *out_height);
/* cv::fisheye::undistortPoints with zero distortion (calls at :93, :322, :333).
 * R and P may be NULL (identity).  pts/out are n (x,y) double pairs. */
VSTAB_API vstab_status vstab_fisheye_undistort_points(const double *pts, int n, const double K[9],
                                                      const double *R, const double *P,
                                                      double *out);
/* The 17 cl_float kernel arguments of createMap in argument order (:283-299):
 * src cx,cy,fx,fy ; out cx,cy,fx,fy ; rot00..rot22 (double -> float casts). */
VSTAB_API void vstab_map_params(const double K_in[9], const double K_out[9], const double R[9],
                                float params[17]);

/* ------------------------------------------------------------------------------------------
 * Stateless device operators (one HIP kernel each; all pointers are DEVICE pointers).
 * ------------------------------------------------------------------------------------------ */

/* Replaces convert_ocl_images_to_nv12_umat, FrameSourceFfmpegOpenCl.cpp:12-93: packs a pitched
 * luma plane (w x h) and a pitched interleaved chroma plane (w/2 x h/2 texels of 2 bytes) into
 * one contiguous (h*3/2) x w buffer.  w and h must be even ("Mismatched image dimensions"). */
VSTAB_API vstab_status vstab_pack_nv12(const void *y, size_t pitch_y, const void *uv,
                                       size_t pitch_uv, int width, int height, void *dst_nv12,
                                       void *stream);

/* Replaces cvtColor(COLOR_YUV2BGR_NV12), FrameSourceWarp.cpp:401.  dst is BGR8, pitch_dst bytes
 * per row (>= 3*width). */
/* 10-bit input (BASELINE config 5): P010 / P016 planes -- 16-bit little-endian samples, significant bits at the top,
 * pitches in bytes -- narrowed to packed 8-bit NV12 (byte = sample >> 8), which the rest of the path works on.
 * No reference counterpart: the reference only accepts 8-bit NV12 (FrameSourceFfmpegOpenCl.cpp:53-56). */
VSTAB_API vstab_status vstab_pack_p010(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                       int width, int height, void *dst_packed_nv12, void *stream);
VSTAB_API vstab_status vstab_cvt_nv12_bgr(const void *y, size_t pitch_y, const void *uv,
                                          size_t pitch_uv, int width, int height, void *dst_bgr,
                                          size_t pitch_dst, void *stream);

/* Replaces the createMap OpenCL kernel, createMap.cl:1-51 + launch at FrameSourceWarp.cpp:275-304.
 * map_x / map_y: float planes, pitch in bytes.  cols, rows <= 32767 (createMap.cl:10-11 uses
 * short indices). */
VSTAB_API vstab_status vstab_create_map(void *map_x, size_t pitch_x, void *map_y, size_t pitch_y,
                                        int cols, int rows, const float params[17], void *stream);

/* Replaces cv::remap(INTER_LINEAR, BORDER_CONSTANT 0), FrameSourceWarp.cpp:306-312, for an
 * 8-bit source of `channels` (1 or 3) interleaved channels. */
VSTAB_API vstab_status vstab_remap_bilinear(const void *src, size_t pitch_src, int src_width,
                                            int src_height, int channels, const void *map_x,
                                            size_t pitch_x, const void *map_y, size_t pitch_y,
                                            void *dst, size_t pitch_dst, int dst_width,
                                            int dst_height, void *stream);

/* The fused hot kernel: cvtColor (:401) + createMap (createMap.cl) + remap (:306-312) in one
 * pass, NV12 in -> BGR8 out, map never written to memory.  Bit-identical to running the three
 * operators above in sequence. */
VSTAB_API vstab_status vstab_warp_nv12_bgr(const void *y, size_t pitch_y, const void *uv,
                                           size_t pitch_uv, int src_width, int src_height,
                                           const float params[17], void *dst_bgr, size_t pitch_dst,
                                           int dst_width, int dst_height, void *stream);

/* The same fused warp with cv::remap's INTER_NEAREST: FrameSourceWarp.hpp:90 takes an InterpolationFlags argument and
 * FrameSourceWarp.cpp:311 hands it to cv::remap; the reference's callers only ever pass INTER_LINEAR.  OpenCV's CPU path:
 * cvRound (half to even) + saturate_cast<short> of each map entry, one source pixel, 0 outside (BORDER_CONSTANT). */
VSTAB_API vstab_status vstab_warp_nv12_nearest(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                               int src_width, int src_height, const float params[17], void *dst_bgr,
                                               size_t pitch_dst, int dst_width, int dst_height, void *stream);
/* The same with the map arithmetic chosen: VSTAB_MAP_CREATEMAP_CL (0, what the call above uses) or VSTAB_MAP_CREATEMAP_CL_OPENCL (5). */
VSTAB_API vstab_status vstab_warp_nv12_nearest_ex(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                                  int src_width, int src_height, const float params[17], int map_mode, void *dst_bgr,
                                                  size_t pitch_dst, int dst_width, int dst_height, void *stream);

/* ------------------------------------------------------------------------------------------
 * the camera surface of the CLI's libdewobble filter (render.ts:611-617,
 * 669-683, 711-717: in_p / out_p in {fish, rect}) and NV12 output for the encoder hand-off
 * (render.ts:275-281).  libdewobble is not part of the reference tree: these modes are defined by this
 * library and have no reference output to compare with.
 * ------------------------------------------------------------------------------------------ */
typedef enum vstab_map_mode {
    VSTAB_MAP_CREATEMAP_CL = 0, /* the reference kernel, quirks included (0/0 on the axis, mirrored rays behind the camera) */
    VSTAB_MAP_FISH_TO_RECT = 1, /* fisheye input -> pinhole output: createMap.cl's arithmetic without those two quirks */
    VSTAB_MAP_FISH_TO_FISH = 2,
    VSTAB_MAP_RECT_TO_RECT = 3,
    VSTAB_MAP_RECT_TO_FISH = 4,
    /* createMap.cl with the arithmetic the reference's OWN kernel has when ROCm's OpenCL compiler builds it for this GPU
     * (oracle/_ref/createMap.gfx950.co: contracted multiply-adds, reciprocal-based division, v_sqrt_f32, ocml's atan --
     * all inside OpenCL 1.2's error bounds, none IEEE-rounded).  Bit-identical to that code object run on the same
     * device (tests/test_refcl_gpu.py) and the arithmetic the pipeline object uses by default
     * (vstab_config.map_precision); mode 0 is the same kernel with every operation IEEE-rounded, reproducible on a
     * CPU.  The two differ in the last bits of the map.  Served wherever mode 0 is: map planes,
     * quantised map, fused warp, per-row (rolling-shutter) warp, nearest-neighbour warp, 10-bit warp. */
    VSTAB_MAP_CREATEMAP_CL_OPENCL = 5
} vstab_map_mode;
typedef enum vstab_out_format {
    VSTAB_OUT_BGR8 = 0, /* what FrameSourceWarp emits (FrameSourceWarp.cpp:313) */
    VSTAB_OUT_NV12 = 1, /* BGR result converted with cvtColor(COLOR_BGR2YUV_I420) arithmetic, chroma interleaved */
    /* PLANE-WISE warp, no colour round trip (SURVEY.md 8(f) row 2 as written: what a filter between NV12 surfaces does,
     * render.ts:606-607, 664-665, 688): the two planes of the source are remapped as they are.
     *   luma    cv::remap(INTER_LINEAR, BORDER_CONSTANT 16) of the Y plane with the map and 1/32-pixel quantisation of the BGR path;
     *   chroma  cv::remap(INTER_LINEAR, BORDER_CONSTANT (128, 128)) of the interleaved UV plane (2 channels, w/2 x h/2).  Output chroma
     *           sample (cx, cy) -- ceil(dst_width / 2) x ceil(dst_height / 2) of them -- sits on luma pixel (2 cx, 2 cy), as source
     *           chroma sample (i, j) sits on source luma pixel (2 i, 2 j): the siting the BGR path's conversions imply (NV12 -> BGR
     *           replicates a chroma sample over its 2 x 2 block, BGR -> NV12 takes the block's top-left pixel).  Its position in the
     *           source chroma plane is that luma pixel's map entry halved, (mapx / 2, mapy / 2) -- exact in fp32 -- and cv::remap
     *           quantises it like any map: cvRound(32 * (map / 2)).  A sample offset common to both frames (MPEG-2's half-pixel
     *           vertical shift) cancels to first order;
     *   border  limited-range black (Y 16, U = V 128): what the BGR path's border, cv::remap's Scalar(0), is in this colour space
     *           (a border of 0 would be green).  As in cv::remap a footprint wholly outside the source gives the border value
     *           and a tap outside it enters the blend as the border value.
     * Not the bytes of VSTAB_OUT_NV12 (which blends converted BGR pixels and converts back); on in-gamut content the two agree within
     * a level.  No reference counterpart: defined here and in the test infrastructure's plain-C statement (vo_remap_plane). */
    VSTAB_OUT_NV12_PLANAR = 2
} vstab_out_format;
/* vstab_create_map with a projection pair.  params as for vstab_create_map (focal lengths in pixels). */
VSTAB_API vstab_status vstab_create_map_ex(void *map_x, size_t pitch_x, void *map_y, size_t pitch_y,
                                           int cols, int rows, const float params[17], int map_mode,
                                           void *stream);
/* vstab_warp_nv12_bgr with a projection pair and an output format.  VSTAB_OUT_NV12 / VSTAB_OUT_NV12_PLANAR: dst is the luma
 * plane (width bytes per row), dst_uv the interleaved chroma plane (ceil(height/2) rows of 2*ceil(width/2) bytes);
 * dst_uv is ignored for VSTAB_OUT_BGR8. */
VSTAB_API vstab_status vstab_warp_nv12_ex(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                          int src_width, int src_height, const float params[17], int map_mode,
                                          int out_format, void *dst, size_t pitch_dst, void *dst_uv,
                                          size_t pitch_dst_uv, int dst_width, int dst_height, void *stream);

/* Rolling-shutter warp (BASELINE.json config 5: "rolling-shutter per-row warp"; no counterpart in the reference, whose gyro
 * path is a stub: gpmf.cpp:5-11): vstab_warp_nv12_ex for map modes 0 / 1 / 5 with a rotation per OUTPUT ROW.  Row y is mapped
 * with the matrix whose nine entries are interpolated in fp32 between params[8..16] (first row) and rot_bottom (last row):
 * t = (float)y / (float)max(dst_height - 1, 1) (IEEE division), m_k = fmaf(t, rot_bottom[k] - params[8 + k], params[8 + k]);
 * the row is then what the mode's createMap arithmetic makes of m -- in mode 5, what the reference's kernel returns for that
 * row when it is handed m as its rotation. */
VSTAB_API vstab_status vstab_warp_nv12_rs(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                          int src_width, int src_height, const float params[17], const float rot_bottom[9],
                                          int map_mode, int out_format, void *dst, size_t pitch_dst, void *dst_uv,
                                          size_t pitch_dst_uv, int dst_width, int dst_height, void *stream);


/* 10-bit pixel path (BASELINE.json config 5: "4K P010, fp16 blend, rolling-shutter per-row warp"; the reference is 8-bit
 * throughout, so this operator is DEFINED here and in the oracle, vo_warp_p010).  y / uv: P010 planes, 16-bit
 * little-endian samples with the 10 significant bits at the top, pitches in bytes, chroma interleaved U,V at half
 * resolution.  Conversion: BT.601 limited range with the cvtColor constants at 10 bits (offsets 64 / 512).  Map: as
 * vstab_warp_nv12_ex (all five modes); rot_bottom != NULL adds vstab_warp_nv12_rs's rotation per output row.  Blend:
 * VSTAB_BLEND_EXACT = the integer four-product sum of the 8-bit path, (sum + 512) >> 10; VSTAB_BLEND_FP16 = four fused
 * multiply-adds in binary16 (weights w / 1024, taps 00, 01, 10, 11), round to nearest even, clamp: within 2 levels of
 * the exact blend.  dst: BGR, three 16-bit samples per pixel, values 0..1023 (low-aligned). */
enum { VSTAB_BLEND_EXACT = 0, VSTAB_BLEND_FP16 = 1 };
VSTAB_API vstab_status vstab_warp_p010(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                       int src_width, int src_height, const float params[17], const float *rot_bottom,
                                       int map_mode, int blend, void *dst_bgr16, size_t pitch_dst, int dst_width,
                                       int dst_height, void *stream);

/* The plane-wise warp at 10 bits: P010 planes in, P010 planes out, no colour round trip -- VSTAB_OUT_NV12_PLANAR's definition on
 * the ten significant bits of each word (sample = word >> 6; border 64 / 512; result word = value << 6).  blend: VSTAB_BLEND_EXACT,
 * (sum + 512) >> 10 per sample, or VSTAB_BLEND_FP16, vstab_warp_p010's binary16 chain per sample.  All map modes; rot_bottom != NULL
 * adds the rotation per output row (modes 0 / 1 / 5).  dst_y: dst_width words per row; dst_uv: ceil(dst_height / 2) rows of
 * ceil(dst_width / 2) (U, V) word pairs.  Pitches in bytes; planes 2-byte aligned, chroma pairs 4-byte aligned. */
VSTAB_API vstab_status vstab_warp_p010_planar(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int src_width, int src_height,
                                              const float params[17], const float *rot_bottom, int map_mode, int blend, void *dst_y,
                                              size_t pitch_dst_y, void *dst_uv, size_t pitch_dst_uv, int dst_width, int dst_height, void *stream);

/* vstab_warp_p010 with P010 planes out in the same kernel (no 16-bit BGR frame in memory): the blended 10-bit BGR pixel is converted
 * in registers by vstab_cvt_bgr16_p010's arithmetic (below).  Served by the LDS-tiled kernel only: VSTAB_ERR_UNSUPPORTED unless the
 * source planes and pitches are 16-byte aligned and the map is VSTAB_MAP_CREATEMAP_CL or VSTAB_MAP_FISH_TO_RECT -- then warp to BGR and convert. */
VSTAB_API vstab_status vstab_warp_p010_planes(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv, int src_width, int src_height,
                                              const float params[17], const float *rot_bottom, int map_mode, int blend, void *dst_y,
                                              size_t pitch_dst_y, void *dst_uv, size_t pitch_dst_uv, int dst_width, int dst_height, void *stream);

/* The 10-bit path's encoder hand-off: BGR as vstab_warp_p010 writes it -> P010 planes (16-bit samples, significant bits at the top;
 * chroma interleaved U, V, ceil(w/2) x ceil(h/2) pairs, from the top-left pixel of each 2 x 2 block).  The BGR -> YUV arithmetic of the
 * NV12 output (cvtColor's BT.601 constants) at 10 bits, offsets 64 / 512; defined in the oracle (vo_cvt_bgr10_p010).  Pitches in bytes. */
VSTAB_API vstab_status vstab_cvt_bgr16_p010(const void *src_bgr16, size_t pitch_src, int width, int height, void *dst_y, size_t pitch_y,
                                            void *dst_uv, size_t pitch_uv, void *stream);

/* ------------------------------------------------------------------------------------------
 * Tracking front-end (device images in; small point lists on the host, as in the reference where
 * goodFeaturesToTrack / calcOpticalFlowPyrLK return std::vector<Point2f>).  These calls
 * synchronise `stream` before returning because their outputs live in host memory.
 * ------------------------------------------------------------------------------------------ */

/* cv::pyrDown (5x5 binomial, REFLECT_101) as used for the LK pyramid; dst is ((w+1)/2, (h+1)/2). */
VSTAB_API vstab_status vstab_pyr_down(const void *src, size_t pitch_src, int width, int height,
                                      void *dst, size_t pitch_dst, void *stream);

/* Two pyramid levels in one launch, as the pipeline builds levels 2 and 3 of the LK pyramid: mid = pyrDown(src) ((w+1)/2 x (h+1)/2),
 * dst = pyrDown(mid) -- the bytes of two vstab_pyr_down calls (the small levels are launch- and latency-bound as kernels of their own). */
VSTAB_API vstab_status vstab_pyr_down_x2(const void *src, size_t pitch_src, int width, int height, void *mid, size_t pitch_mid,
                                         void *dst, size_t pitch_dst, void *stream);

/* cornerMinEigenVal(blockSize 3, ksize 3): dense width x height float response (device). */
VSTAB_API vstab_status vstab_min_eig(const void *gray, size_t pitch, int width, int height,
                                     void *eig_f32, void *stream);

/* Replaces find_corners -> goodFeaturesToTrack(gray, max_corners, quality, min_distance),
 * FrameSourceWarp.cpp:228-240 (the reference passes 200, 0.01, 30).  xy receives up to
 * max_corners (x,y) float pairs in acceptance order; *count the number found. */
VSTAB_API vstab_status vstab_good_features(const void *gray, size_t pitch, int width, int height,
                                           int max_corners, double quality, double min_distance,
                                           float *xy, int *count, void *stream);

/* The same with a choice of detector.  AUTO: one fused pass (eigenvalue, threshold and 3x3 maximum test in one kernel,
 * the eigenvalue map is never stored); a frame with more than 2^18 corners above the threshold (an eigenvalue plateau
 * over most of a large frame) falls back to TWO_PASS, which stores the map (vstab_min_eig), scans it and grows its key
 * buffer as needed.  Both give the same corners.  *detector_used (may be NULL) receives VSTAB_DETECTOR_FUSED or
 * VSTAB_DETECTOR_TWO_PASS. */
enum { VSTAB_DETECTOR_AUTO = 0, VSTAB_DETECTOR_TWO_PASS = 1, VSTAB_DETECTOR_FUSED = 2 };
VSTAB_API vstab_status vstab_good_features_ex(const void *gray, size_t pitch, int width, int height,
                                              int max_corners, double quality, double min_distance, int detector,
                                              float *xy, int *count, int *detector_used, void *stream);

/* Replaces calcOpticalFlowPyrLK with default parameters (win 21x21, maxLevel 3, 30 iterations,
 * eps 0.01, minEigThreshold 1e-4), FrameSourceWarp.cpp:252.  prev/next: device gray images of the
 * same size; prev_xy: n host (x,y) pairs; next_xy / status: host outputs (n pairs / n bytes). */
VSTAB_API vstab_status vstab_pyr_lk(const void *prev, size_t pitch_prev, const void *next,
                                    size_t pitch_next, int width, int height, const float *prev_xy,
                                    int n, float *next_xy, unsigned char *status, void *stream);

/* ------------------------------------------------------------------------------------------
 * Host-side motion model.
 * ------------------------------------------------------------------------------------------ */

/* Replaces guess_camera_rotation, FrameSourceWarp.cpp:316-375: undistort both point sets, random
 * depths (seeded PCG32 instead of the reference's un-seeded rand()), PnP-RANSAC (100 iterations,
 * 8 px, 0.99), Rodrigues.  prev_xy/cur_xy: n (x,y) float pairs in input pixels.  R: row-major
 * rotation since the last frame; *inliers: RANSAC inlier count (the reference's return value). */
VSTAB_API vstab_status vstab_estimate_rotation(const float *prev_xy, const float *cur_xy, int n,
                                               const double K_in[9], const double K_out[9],
                                               uint64_t seed, double R[9], int *inliers);

/* Savitzky-Golay weights of gram_sg::SavitzkyGolayFilterConfig(m, 0, 2, 0) (:212): 2m+1 doubles. */
VSTAB_API vstab_status vstab_sg_weights(int m, double *weights);

/* gram_sg::RotationFilter replacement (:212,444,459,471): ring of 2m+1 matrices that starts
 * zero-filled; filter() = polar factor U*V^T of the weighted sum. */
typedef struct vstab_rotation_filter vstab_rotation_filter;
VSTAB_API vstab_status vstab_rotation_filter_create(int m, vstab_rotation_filter **out);
VSTAB_API vstab_status vstab_rotation_filter_add(vstab_rotation_filter *f, const double R[9]);
VSTAB_API vstab_status vstab_rotation_filter_filter(const vstab_rotation_filter *f, double R_out[9]);
VSTAB_API void vstab_rotation_filter_destroy(vstab_rotation_filter *f);

/* Gyro samples -> the per-frame rotations the pipeline takes from an external sensor (vstab_frame.delta_rotation,
 * vstab_frame.readout_rotation): the step the reference stubbed (gpmf.cpp:5-11 declares GyroFrame {start_ts, end_ts, roll,
 * pitch, yaw} and the commented-out reader fills one per GPMF "GYRO" sample; AvFrameSourceFileVaapi.cpp:121-123 "TODO process
 * GPMF packet").  vstab_gyro_sample IS that record: an angular rate held from start_ts to end_ts (seconds on the clock the
 * frame timestamps use), about the camera's z (roll), x (pitch) and y (yaw) axes, x right, y down, z along the optical axis.
 * Over an interval [a, b] the rotation is the ordered product of exponential maps, later samples on the left (the
 * accumulation order of FrameSourceWarp.cpp:441):
 *     R(a, b) = exp([w_n] dt_n) ... exp([w_1] dt_1),   w_i = rate_scale * (pitch_i, yaw_i, roll_i),
 *     dt_i = length of [start_ts_i, end_ts_i] inside [a, b]   (time not covered by any sample contributes nothing).
 * rate_scale = -1 for a gyro that reports the rate of the camera BODY in rad/s (points of the scene then move by R in camera
 * coordinates, which is what guess_camera_rotation returns); it also absorbs unit conversion (-pi/180 for deg/s).
 * R_delta = R(t_prev_first_row, t_first_row): the camera's rotation since the previous frame; R_readout =
 * R(t_first_row, t_last_row): its rotation during this frame's read-out (rolling shutter).  Either may be NULL.  Samples
 * must be ordered by start_ts with end_ts >= start_ts.  fp64, host only.  GPMF container parsing (gpmf-parser) stays upstream. */
typedef struct vstab_gyro_sample {
    double start_ts, end_ts, roll, pitch, yaw; /* field order of the reference's GyroFrame */
} vstab_gyro_sample;
VSTAB_API vstab_status vstab_gyro_integrate(const vstab_gyro_sample *samples, int n, double rate_scale,
                                            double t_prev_first_row, double t_first_row, double t_last_row,
                                            double R_delta[9], double R_readout[9]);

/* GPMF payload -> gyro samples: the reader the reference sketched and left commented out (opencv/gpmf.cpp:33-114, with gpmf-parser;
 * AvFrameSourceFileVaapi.cpp:121-123 "TODO process GPMF packet").  payload / n: one packet of the camera's metadata stream as the
 * demuxer hands it over; pkt_ts / pkt_dur: the packet's time stamp and duration in seconds (AVPacket.pts / .duration times the
 * stream's time base).  GoPro's published KLV layout is walked directly (no gpmf-parser): items of {FourCC key, type character,
 * structure size (1 byte), repeat count (2 bytes)} + repeat * size bytes, everything big-endian and padded to four bytes; type 0
 * nests (DEVC > STRM > items).  Inside a stream, SCAL holds the divisor -- one for all elements or one per element -- that turns
 * the raw GYRO integers (type 's', three per sample; 'S' 'l' 'L' 'b' 'B' 'f' 'd' are accepted too) into rad/s.  As gpmf.cpp:95-101
 * intends, a block's samples share the packet's span evenly -- sample i of m: start_ts = pkt_ts + pkt_dur * i / m, end_ts = start_ts
 * + pkt_dur / m -- and elements 0, 1, 2 go to roll, pitch, yaw: the order GoPro documents for HERO5 and later, Z (the optical axis),
 * X (to the right), Y (downwards), which is vstab_gyro_sample's convention.  The result feeds vstab_gyro_integrate (rate_scale = -1
 * for these body rates).  Up to `cap` samples are written to out; *n_out = the number found (call again with a larger buffer if it
 * exceeds cap).  Every GYRO block of the payload is taken, in order.  VSTAB_ERR_INVALID for a payload that is not well formed: an
 * item that runs past its container, a truncated header, nesting deeper than eight levels, a GYRO block that does not have three
 * elements per sample, a SCAL of zero or not finite, a packet time stamp or duration that is not finite.  Host only; reads nothing
 * outside [payload, payload + n). */
VSTAB_API vstab_status vstab_gpmf_parse_gyro(const void *payload, size_t n, double pkt_ts, double pkt_dur, vstab_gyro_sample *out, int cap,
                                             int *n_out);

/* A map that does not change between frames (tracking off: undistort only, the CLI's stab=none re-projections) need
 * not be evaluated per frame as the reference does (FrameSourceWarp.cpp:283-304): vstab_quantised_map writes, once,
 * what cv::remap makes of every map entry (32 * map rounded to int; vstab_quantised_map_bytes() bytes, 16-byte aligned
 * device memory), and vstab_warp_nv12_mapped warps with it -- same integers, same pixels as vstab_warp_nv12_ex. */
VSTAB_API size_t vstab_quantised_map_bytes(int dst_width, int dst_height);
VSTAB_API vstab_status vstab_quantised_map(void *qmap, int dst_width, int dst_height, const float params[17],
                                           int map_mode, void *stream);
VSTAB_API vstab_status vstab_warp_nv12_mapped(const void *y, size_t pitch_y, const void *uv, size_t pitch_uv,
                                              int src_width, int src_height, const void *qmap, int out_format,
                                              void *dst, size_t pitch_dst, void *dst_uv, size_t pitch_dst_uv,
                                              int dst_width, int dst_height, void *stream);

/* The `debug` overlay of the filter surface (render.ts:678): a filled (2*half+1)^2 square of colour bgr (0x00RRGGBB;
 * the low byte alone for a 1-channel plane) at each of n centres (x, y int pairs in DEVICE memory), clipped. */
VSTAB_API vstab_status vstab_draw_markers(void *dst, size_t pitch, int width, int height, int channels,
                                          const int *centres_xy_device, int n, int half, unsigned int bgr,
                                          void *stream);

/* ------------------------------------------------------------------------------------------
 * The pipeline object: drop-in for FrameSourceWarp behind the FrameSource pull interface
 * (FrameSource.hpp:9-24, FrameSourceWarp.hpp:83-91).
 * ------------------------------------------------------------------------------------------ */

/* One NV12 frame handed over by the upstream source (what FrameSourceFfmpegOpenCl produces,
 * FrameSourceFfmpegOpenCl.cpp:58-85).  A packed (h*3/2 x w) buffer is uv = y + pitch_y*height.
 * By default the planes only need to stay valid until the callback is called again (see `hold`) or the
 * handle is destroyed: the library copies them into its own HBM ring. */
typedef struct vstab_frame {
    const void *y;
    const void *uv;
    size_t pitch_y, pitch_uv;
    int width, height; /* luma size; both even */
    int mem;           /* 0 = device memory, 1 = host memory, 2 (VSTAB_MEM_DMABUF) = a DMA-BUF: see dmabuf_fd below */
    int64_t pts;
    const double *delta_rotation; /* optional (NULL = none): 3x3 row-major rotation of the camera since the previous
                          frame from an external sensor -- the gyro path the reference stubs (gpmf.cpp:5-11,
                          AvFrameSourceFileVaapi.cpp:121-123).  Used in place of the optical-flow
                          estimate (guess_camera_rotation's return value, FrameSourceWarp.cpp:429) when
                          vstab_config.tracking == 0; read during the callback only. */
    int bit_depth;     /* 0 or 8: 8-bit NV12.  10 / 12 / 16: P010-style planes (16-bit little-endian samples, significant
                          bits at the top, pitches in bytes), narrowed to 8 bits on ingest (vstab_pack_p010). */
    int hold;          /* how many FURTHER pull callbacks these planes stay valid and unchanged for.  0 (default): only
                          until the next callback -- the library then waits for its copy of this frame to finish before
                          it calls upstream again (a decoder that recycles one output surface).  Ref-counted or pooled
                          frames can say how deep the pool is and the wait disappears from the frame loop.  From
                          smooth_radius + read_ahead + 6 on (smooth_radius + 18 with the default read-ahead of twelve), the
                          planes are not copied at all but read in place by the tracker and
                          by the warp (which runs on vstab_config.stream): the callback at which the promise runs out
                          first waits, on the host, for that warp to finish.  1 << 29 or more: never waited for.
                          That threshold is the 8-BIT rule.  16-bit (P010) frames are always narrowed into library
                          memory for the tracker; their own planes are read in place by the 10-bit warp only with
                          hold >= 1 << 29 (the promise has to cover a warp that runs smooth_radius frames later, and
                          finite promises are not tracked for them) -- anything less and they are copied on ingest. */
    const double *readout_rotation; /* optional (NULL = global shutter): 3x3 row-major rotation of the camera between the
                          exposure of this frame's first and last row (rolling shutter, from the same sensor as
                          delta_rotation; BASELINE.json config 5).  The frame is then warped with a rotation per output row:
                          the stabilising rotation W for the first row, readout_rotation * W for the last, matrix entries
                          interpolated in between (vstab_warp_nv12_rs; preset and fisheye -> rectilinear maps only).  Read
                          during the callback only. */
    int dmabuf_fd;     /* mem == VSTAB_MEM_DMABUF: the frame lives in a DMA-BUF object -- what a VAAPI / AMF decoder surface is
                          once exported (av_hwframe_map(..., AV_PIX_FMT_DRM_PRIME) -> AVDRMFrameDescriptor: objects[0].fd /
                          .size, layers[].planes[].offset / .pitch).  `y` and `uv` are then BYTE OFFSETS of the two planes
                          inside the object (cast to pointers), pitch_y / pitch_uv as usual.  The library imports the object
                          into the HIP address space (hipImportExternalMemory; cached per object, the fd is not consumed and
                          may be closed after the callback) and uses the planes as device memory under the `hold` rules above:
                          the zero-copy replacement of the reference's VAAPI -> host -> OpenCL double copy
                          (AvFrameSourceMapOpenCl.cpp:17-66).  Ignored for other `mem` values. */
    size_t dmabuf_size; /* size of the object in bytes (AVDRMObjectDescriptor.size) */
    uint64_t dmabuf_modifier; /* AVDRMObjectDescriptor.format_modifier of the object: how its bytes are arranged.  The kernels read
                          rows of `pitch` bytes, so only DRM_FORMAT_MOD_LINEAR (0) is accepted -- and DRM_FORMAT_MOD_INVALID
                          (0x00ffffffffffffff: "no modifier given", which libav reports for implicit-layout exports; such a surface
                          is the exporter's to keep linear).  Any other value (a tiled / compressed surface) is refused with
                          VSTAB_ERR_UNSUPPORTED instead of being read as garbage: map the surface linear (hwmap) first. */
} vstab_frame;
#define VSTAB_DRM_FORMAT_MOD_LINEAR 0ull
#define VSTAB_DRM_FORMAT_MOD_INVALID 0x00ffffffffffffffull
enum { VSTAB_MEM_DEVICE = 0, VSTAB_MEM_HOST = 1, VSTAB_MEM_DMABUF = 2 };

/* Upstream FrameSource (FrameSource.hpp:14,20): return 0 and fill *out, VSTAB_EOF (-1) at end of
 * stream, any other value on failure (propagated as VSTAB_ERR_SOURCE, like a rethrown int). */
typedef struct vstab_source {
    int (*pull)(void *user, vstab_frame *out);
    int (*peek)(void *user, vstab_frame *out);
    void *user;
} vstab_source;

/* SG = the reference (gram_sg, FrameSourceWarp.cpp:212); NONE = no correction (libdewobble stab=none with tracking on);
 * FIXED = hold the first frame's orientation (libdewobble stab=fixed, render.ts:676); KALMAN = opencv/kalman constants. */
enum { VSTAB_SMOOTHER_SG = 0, VSTAB_SMOOTHER_KALMAN = 1, VSTAB_SMOOTHER_NONE = 2, VSTAB_SMOOTHER_FIXED = 3 };

/* Lens description of the CLI's libdewobble filter (render.ts:611-617,669-683): projection + diagonal field of view. */
typedef enum vstab_projection { VSTAB_PROJ_RECT = 0, VSTAB_PROJ_FISH = 1 } vstab_projection;
/* Camera matrix of such a lens: f = (d/2)/tan(dfov/2) (rect) or (d/2)/(dfov/2) (fish, equidistant), d = the image
 * diagonal in pixels; principal point (cx, cy), negative = the image centre (width/2, height/2) as render.ts:682-683. */
VSTAB_API vstab_status vstab_lens_camera(int projection, double dfov_deg, int width, int height, double cx,
                                         double cy, double K[9]);

/* Constructor arguments of FrameSourceWarp (FrameSourceWarp.hpp:83-91) plus what the reference
 * hard-codes.  vstab_config_default fills the reference's defaults. */
typedef struct vstab_config {
    int abi_version;     /* VSTAB_ABI_VERSION, written by vstab_config_default; vstab_create refuses anything else */
    int preset;          /* vstab_camera_preset */
    double scale;        /* 1 */
    int crop_borders;    /* 0 */
    double zoom;         /* 1 */
    int smooth_radius;   /* 30.  >= 1 is the reference's filter; 0 (where gram_sg's weights divide by zero) is defined here as
                            "no smoothing": the single weight is 1 and every frame is warped by the identity correction */
    int interpolation;   /* cv::InterpolationFlags: 1 = INTER_LINEAR (the only mode the reference ever passes), 0 = INTER_NEAREST
                            (lens_mode 0, 8-bit BGR output, no read-out rotations); others are refused */
    int smoother;        /* VSTAB_SMOOTHER_SG (reference behaviour) */
    int tracking;        /* 1; 0 = no optical flow: rotations are identity (BASELINE config 1: undistort only) or the
                            upstream-supplied vstab_frame.delta_rotation (external gyro), smoothed the same way */
    uint64_t seed;       /* PCG32 seed replacing the reference's un-seeded rand() */
    void *stream;        /* hipStream_t the warp (and so dst) is enqueued on; NULL = default stream.  Ingest and
                            tracking run on internal streams that overlap it; ordering is by events.  The runtime has
                            four hardware queues per process: beside a NULL stream the handle creates three streams
                            (tracking, read-ahead, speculative corner detection), beside a stream of the caller's two
                            (the detection then queues on the read-ahead stream; VSTAB_DETECT_STREAM=1 in the
                            environment gives it its own) -- INTEGRATION.md section 3. */
    /* libdewobble-style lens surface.  lens_mode 0 (default) = the reference's preset cameras
     * and createMap.cl; 1 = the fields below replace preset / scale / crop_borders / zoom. */
    int lens_mode;
    int in_projection;   /* in_p: vstab_projection */
    int out_projection;  /* out_p */
    double in_dfov;      /* in_dfov, degrees, (0, 360) for fish and (0, 180) for rect */
    double out_dfov;     /* out_dfov; 0 = in_dfov (render.ts:673) */
    int out_width;       /* out_w; 0 = input width (render.ts:678) */
    int out_height;      /* out_h; 0 = input height */
    double out_cx;       /* out_fx "focal point"; negative = out_width / 2 (render.ts:682) */
    double out_cy;       /* out_fy; negative = out_height / 2 */
    int debug;           /* debug (render.ts:678): mark the features tracked into each emitted frame (green 7x7 squares;
                            luma 235 in NV12 output) at the positions the warp sends them to.  Works in both lens modes. */
    int pixel_depth;     /* 8 (default, 0 = 8): the reference's 8-bit path; P010 input is narrowed on ingest.  10 (BASELINE.json
                            config 5): upstream must hand P010 device frames (vstab_frame.bit_depth > 8); the tracker runs on
                            the narrowed luma exactly as in the 8-bit path, the frame is warped from the 16-bit planes with
                            vstab_warp_p010 and emitted by vstab_pull_frame_bgr16. */
    int blend;           /* pixel_depth 10: VSTAB_BLEND_EXACT (default) or VSTAB_BLEND_FP16 */
    int read_ahead;      /* frames the library pulls from upstream AHEAD of the frame it returns, beyond smooth_radius: 0 = the default
                            (12), else 1 .. 16.  The read-ahead is what lets copy, pyramid, corner detection and tracking overlap the
                            host's work; every output is identical for any value.  A live source sees a latency of
                            smooth_radius + read_ahead + 2 frames (the reference: smooth_radius) -- a capture pipeline that cares
                            lowers it (1: 30 + 3 frames at the reference's radius) and pays in frames/s. */
    int map_precision;   /* lens_mode 0: VSTAB_MAP_PRECISION_OPENCL (default; VSTAB_MAP_CREATEMAP_CL_OPENCL: the arithmetic the reference's
                            own kernel -- createMap.cl through ROCm's OpenCL compiler -- has on this GPU, bit-identical to it) or
                            VSTAB_MAP_PRECISION_IEEE (createMap.cl with every operation IEEE-rounded: reproducible by a CPU, a few
                            output bytes per frame away from the reference's GPU result).  Every path of the
                            handle honours it: cached map (tracking off), read-out rotations, INTER_NEAREST, pixel_depth 10.
                            lens_mode 1 ignores it (those maps are this library's own definitions, IEEE throughout). */
} vstab_config;
enum { VSTAB_MAP_PRECISION_IEEE = 0, VSTAB_MAP_PRECISION_OPENCL = 1 };

typedef struct vstab_handle vstab_handle;

VSTAB_API void vstab_config_default(vstab_config *cfg);
/* FrameSourceWarp::FrameSourceWarp (:199-226): peeks the first upstream frame for the size,
 * derives both cameras, allocates the look-ahead ring in HBM. */
VSTAB_API vstab_status vstab_create(const vstab_config *cfg, const vstab_source *src, vstab_handle **out);
VSTAB_API vstab_status vstab_get_output_info(const vstab_handle *h, int *width, int *height,
                                             double K_in[9], double K_out[9]);
/* FrameSourceWarp::pull_frame (:452-476): consumes upstream frames until smooth_radius+1 are
 * buffered (or EOF), then warps the oldest buffered frame into dst (device BGR8).  Returns
 * VSTAB_EOF when the stream is drained.  The first input frame is never emitted (:403-407).
 * dst is complete once vstab_config.stream is synchronised.  Upstream frames handed to the callbacks
 * must already be complete in memory when the callback returns (they are read on an internal stream).
 * Read-ahead: to overlap copy, pyramid, corner detection and tracking with the host work, the library pulls
 * upstream up to vstab_config.read_ahead + 2 (by default fourteen) frames earlier than the reference's loop would (same frames,
 * same order, same outputs). */
VSTAB_API vstab_status vstab_pull_frame(vstab_handle *h, void *dst_bgr, size_t pitch_dst);
/* The consumer's loop (DisplayImage.cpp:60-70: `while (true) { frame = source.pull_frame(); ... }`) as one call: n consecutive
 * vstab_pull_frame calls, frame i into dst[(first + i) % n_dst] with pitch[(first + i) % n_dst] -- an encoder's ring of output
 * surfaces.  Stops at the first call that does not return VSTAB_OK and returns its status (VSTAB_EOF at end of stream);
 * *n_done (may be NULL) = frames emitted.  Nothing else differs from calling vstab_pull_frame n times. */
VSTAB_API vstab_status vstab_pull_frames(vstab_handle *h, int n, void *const *dst_bgr, const size_t *pitch_dst, int n_dst, int first,
                                         int *n_done);
/* pull_frame with NV12 output for the encoder hand-off (render.ts:275-281): the same frame,
 * converted as vstab_warp_nv12_ex(VSTAB_OUT_NV12) defines.  dst_y: width bytes per row; dst_uv: ceil(height/2) rows
 * of 2*ceil(width/2) bytes.  BGR and NV12 pulls may be mixed freely on one handle. */
VSTAB_API vstab_status vstab_pull_frame_nv12(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv,
                                             size_t pitch_uv);
/* The same pull with the frame warped PLANE BY PLANE -- no colour round trip (VSTAB_OUT_NV12_PLANAR: the Y and UV planes of the
 * source remapped as they are, limited-range black outside; SURVEY.md 8(f) row 2, the route the CLI's filter takes between NV12
 * surfaces, render.ts:606-607, 664-665, 688): fewer instructions per pixel than any route through BGR, 23.0 MB instead of 33.6 MB
 * moved per 4K frame.  Same look-ahead, same rotations, same planes' shapes as vstab_pull_frame_nv12; the map is evaluated for every
 * frame (the quantised-map cache holds no chroma positions).  May be mixed with the other 8-bit pulls on one handle. */
VSTAB_API vstab_status vstab_pull_frame_nv12_planar(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv);
/* pull_frame into HOST memory (what a cv::Mat / imshow consumer of DisplayImage.cpp:63-65 needs): the frame is warped
 * into a buffer of the handle and copied out; returns when the copy has completed. */
VSTAB_API vstab_status vstab_pull_frame_host(vstab_handle *h, void *dst_bgr_host, size_t pitch_dst);
/* pull_frame of a handle created with pixel_depth = 10: device BGR, three 16-bit samples per pixel (values 0..1023),
 * pitch_dst in bytes, >= 6 * width and even.  The 8-bit pull functions refuse such a handle and vice versa. */
VSTAB_API vstab_status vstab_pull_frame_bgr16(vstab_handle *h, void *dst_bgr16, size_t pitch_dst);
/* pull_frame of a pixel_depth = 10 handle as P010 planes for a 10-bit encoder (the counterpart of vstab_pull_frame_nv12): the frame is
 * warped and converted in one kernel (vstab_warp_p010_planes) when the frame's planes allow it, else warped into a 16-bit BGR buffer of the
 * handle and converted by vstab_cvt_bgr16_p010. */
VSTAB_API vstab_status vstab_pull_frame_p010(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv);
/* pull_frame of a pixel_depth = 10 handle warped plane by plane (vstab_warp_p010_planar): P010 planes in, P010 planes out, no
 * conversion to BGR and back; vstab_config.blend selects the blend of each sample. */
VSTAB_API vstab_status vstab_pull_frame_p010_planar(vstab_handle *h, void *dst_y, size_t pitch_y, void *dst_uv, size_t pitch_uv);
/* FrameSourceWarp::peek_frame (:478-480) IS pull_frame in the reference (destructive); kept. */
VSTAB_API vstab_status vstab_peek_frame(vstab_handle *h, void *dst_bgr, size_t pitch_dst);
VSTAB_API void vstab_destroy(vstab_handle *h);

/* Introspection for parity tests and profiling: what consume_frame (:397-450) decided for the
 * index-th consumed frame that produced a rotation (index 0 = second input frame). */
typedef struct vstab_frame_log {
    int key_frame;       /* corners re-detected before tracking this frame (:415-419) */
    int n_corners;       /* corners fed to the tracker */
    int n_tracked;       /* pairs with status != 0 (:263-268) */
    int n_inliers;       /* guess_camera_rotation's return value */
    int fallback;        /* 1 if n_inliers < 40 and the previous rotation / identity was reused (:432-438) */
    double R_frame[9];   /* rotation since last frame actually used */
    double R_accum[9];   /* accumulated measured rotation (:441-442) */
} vstab_frame_log;
/* Indices are absolute (frames since the start); only the most recent 65536 entries are retained. */
VSTAB_API int vstab_frame_log_count(const vstab_handle *h);
VSTAB_API vstab_status vstab_get_frame_log(const vstab_handle *h, int index, vstab_frame_log *out);
/* The rotation handed to the warp for the index-th emitted frame (rotation_correction.inv(), :475). */
VSTAB_API vstab_status vstab_get_warp_rotation(const vstab_handle *h, int index, double R[9]);

/* Per-stage profiler: the role of the reference's Profiler / FrameSourceProfile decorators
 * (Profiler.cpp:14-35), with GPU stages timed by HIP events on the handle's stream (microsecond
 * resolution instead of the reference's whole-millisecond truncation).
 * GPU stages exclude host waits; host stages are steady_clock wall time. */
typedef struct vstab_profile {
    long frames_consumed, frames_emitted, key_frames;
    double gpu_ingest_ms, gpu_pyramid_ms, gpu_corners_ms, gpu_lk_ms, gpu_warp_ms; /* sums of kernel time */
    double host_corners_ms, host_track_wait_ms, host_estimate_ms, host_smooth_ms;  /* sums of wall time */
    long warp_launches;
    long warp_timed;     /* warp launches that gpu_warp_ms sums over (level 1 samples every 8th) */
    long dmabuf_imports, dmabuf_evictions, dmabuf_cached; /* VSTAB_MEM_DMABUF: objects imported so far, unmapped again (least recently
                            used, once more than 256 are cached and none of the window's frames can still refer to them), mapped now */
    long corner_selections_by_caller, corner_selections_by_helper; /* speculative corner detections (planned key frames) whose corners were
                            selected by the calling thread itself because the helper thread had not woken up / by the helper thread */
    long epochs_in_turn; /* planned key frames whose detection and tracker launches ran on the second of the handle's two epoch streams, beside
                            the epoch still being tracked on the first (frames up to 1920 x 1200 with a caller on the default stream; 0 otherwise) */
} vstab_profile;
/* Loads the library's five GPU code objects now.  The HIP runtime loads a code object at the first launch of one of its kernels -- tens of
 * milliseconds in the middle of the first frames -- and on ROCm 7.2 such a late load can FAULT ("write access to a read-only page") when the
 * process has unloaded another module before it (hipModuleUnload; an OpenCL program released by a filter next door): the new code object may
 * be placed where the old one was still mapped read-only.  vstab_create calls this itself; a host that uses the stateless operators
 * (vstab_warp_nv12, vstab_pyr_lk, ...) without a handle, or wants the load at start-up, calls it once after selecting the device. */
VSTAB_API vstab_status vstab_preload_kernels(void);
/* level 0 = off, 1 = time every 8th warp launch only (event records are expensive host calls), 2 = every GPU stage */
VSTAB_API vstab_status vstab_enable_profiling(vstab_handle *h, int level);
/* Synchronises the stream, folds all pending event pairs into the sums and returns them. */
VSTAB_API vstab_status vstab_get_profile(vstab_handle *h, vstab_profile *out);

/* Profiling aid for the stateless warp operators (vstab_warp_nv12_*, vstab_warp_p010): the NEXT such call on this thread
 * launches its kernel with hipExtLaunchKernelGGL and the two events (hipEvent_t with timing enabled, owned by the caller),
 * which then hold the kernel's own start and end -- the duration rocprofv3's kernel trace reports, without launch gaps.
 * One-shot; a call that launches no such kernel leaves the request pending. */
VSTAB_API void vstab_time_next_launch(void *start_event, void *stop_event);

/* Utility upstream source for benchmarks and tests: cycles over n_frames caller-owned device
 * frames (packed NV12, same size) for total_frames pulls, then reports EOF.  Plays the role of
 * the decode chain upstream of FrameSourceWarp (DisplayImage.cpp:42-53), which is out of scope. */
typedef struct vstab_ring_source vstab_ring_source;
VSTAB_API vstab_status vstab_ring_source_create(const void *const *frames, int n_frames, int width, int height,
                                                size_t pitch, long total_frames, vstab_ring_source **out,
                                                vstab_source *as_source);
/* The same for P010-style frames (bit_depth 10 / 12 / 16: 16-bit samples, pitch in bytes, chroma plane at
 * frame + pitch * height) and/or with a read-out rotation per ring frame (9 doubles each, copied; may be NULL). */
VSTAB_API vstab_status vstab_ring_source_create_ex(const void *const *frames, int n_frames, int width, int height,
                                                   size_t pitch, long total_frames, int bit_depth,
                                                   const double *readout_rotations, vstab_ring_source **out,
                                                   vstab_source *as_source);
/* vstab_frame.hold the source reports for every frame.  Default 1 << 30: the frames belong to the caller for the life of the
 * source and are used in place, nothing is copied.  0 = the contract of a decoder that recycles its output surface: every
 * frame is copied into the library's ring (vstab_pack_nv12) before the next callback. */
VSTAB_API void vstab_ring_source_set_hold(vstab_ring_source *s, int hold);
VSTAB_API void vstab_ring_source_destroy(vstab_ring_source *s);

#ifdef __cplusplus
}
#endif
#endif /* VSTAB_H_ */
