/*
 * lab4d_ingest.h -- batch ingestion on the device (included by lab4d_hip.h).
 *
 * Replaces the per-step CPU work of the reference's dataloader workers: `VidDataset.read_raw` (dataloader/vidloader.py:217-262:
 * read_rgb :264-281, read_mask :283-309, read_depth :311-324, read_feature :326-340, read_flow :342-358) with
 * `numpy_utils.bilinear_interp` (utils/numpy_utils.py:97-122), i.e. the gather of `pixels_per_image` random pixels per frame out
 * of the memory-mapped per-video arrays, the pinned-memory collate and the host-to-device copy.  Here the frame data of a video
 * stays RESIDENT IN HBM in its on-disk dtypes (288 GB holds thousands of 256x256 frames) and one launch gathers a whole batch.
 *
 * frame_ptrs: (M,5) int64 DEVICE ADDRESSES of the M frames' slices, built by the host from the cache's base addresses (pure
 *   integer work on M <= a few hundred rows): [rgb (H,W,3) | mask (H,W,2) u8: {segmentation, vis2d} | depth (H,W) |
 *   flow (H,W,3): the FlowFW_d / FlowBW_d table and row vidloader.py:352-355 selects | feature (FR,FR,FC)].
 * xy: (M,N,2) int32 pixel coordinates (x, y) -- `sample_xy` draws them (vidloader.py:183-196).
 * dtype codes: LAB4D_F16 / LAB4D_F32 of the cached arrays (the reference stores rgb, depth, flow as float16: preprocess/libs/io.py:154-159,243).
 * Outputs keep the reference's dtypes and shapes: rgb (M,N,3) and depth (M,N,1) in the cache dtype, mask / vis2d (M,N,1) u8 (bool),
 *   flow (M,N,2) + flow_uct (M,N,1) fp32 (`flow.astype(np.float32)`), feature (M,N,FC) fp32 = f32(cache dtype(bilinear in fp64))
 *   exactly as numpy evaluates numpy_utils.py:106-121 (coordinates rand_xy / H * FR, floor, clip to [0, FR-2], four taps summed
 *   in fp64 in the reference's order, rounded to the feature map's dtype, then to fp32), hxy (M,N,3) fp32 = (x, y, 1).
 * All results are BIT-EXACT with the reference (pure data movement + one fp64 interpolation with the same operation order).
 */
#ifndef LAB4D_INGEST_H
#define LAB4D_INGEST_H

int lab4d_ingest_gather(const int64_t* frame_ptrs, const int32_t* xy, int M, int N, int H, int W, int FR, int FC, int img_dtype,
                        int flow_dtype, int feat_dtype, void* rgb, uint8_t* mask, uint8_t* vis2d, void* depth, float* flow,
                        float* flow_uct, float* feature, float* hxy, void* stream);

#endif /* LAB4D_INGEST_H */
