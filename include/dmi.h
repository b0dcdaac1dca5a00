/*
 * dmi.h -- C ABI of the MI355X-native depth-map-integration (TSDF fusion) path.
 *
 * This is the drop-in boundary for ONE path of bastienjacquet/CudaDepthMapIntegration:
 * the two free functions that vtkCudaReconstructionFilter forward-declares and calls
 * (Reconstruction/vtkCudaReconstructionFilter.cxx:65-71, :171-176), defined in
 * Reconstruction/CudaReconstruction.cu:
 *
 *     void CudaInitialize(vtkMatrix4x4*, int dims[3], double orig[3], double spacing[3],
 *                         double thick, double rho, double eta, double delta, int depthDims[2]);   cu:269-298
 *     template<class T> bool ProcessDepthMap(std::vector<std::string> vti, std::vector<std::string> krtd,
 *                         double thresholdBestCost, vtkDoubleArray* io_scalar);                     cu:302-386
 *
 * Those take VTK objects and file names and do disk I/O inside the GPU loop.  Here the seam
 * is split: the host side (VTK or the VTK-free mirror in cudadepthmapintegration_amd/csrc/host)
 * loads or generates the views; this library only sees plain pointers and sizes.
 *
 * Conventions (all taken from the reference):
 *   - the voxel grid is the CELL grid of the filter's input vtkImageData: cell_dims = point
 *     dims - 1 (filt.cxx:123-124, cu:128-133, cu:330-331); linear voxel id = (k*ny + j)*nx + i,
 *     x fastest (cu:126-134) = vtk cell-id order of the "reconstruction_scalar" array (filt.cxx:129-135);
 *   - 4x4 matrices are row-major as vtkMatrix4x4 / cu:220-230; only rows 0..2 are used (cu:88-93);
 *   - a depth table is W*H values in vtkImageData point order: row 0 is the BOTTOM image row
 *     (cu:141-149); the value -1 means "no depth" (cu:202; Sources/ReconstructionData.cxx:164);
 *   - arithmetic is IEEE fp64 in the reference's expression order, every multiply and add rounded
 *     separately; results are bit-identical to oracle/tsdf_oracle.c on one GPU with an f64 grid.
 *
 * Every function returns DMI_OK (0) or a dmi_status error code and never calls exit()
 * (the reference's gpuAssert does, cu:68-76).  dmi_last_error() gives the message.
 * A context is not thread-safe; distinct contexts may be used from distinct threads
 * (the reference keeps global __constant__ state, cu:55-64, and is not re-entrant).
 */
#ifndef DMI_H_
#define DMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMI_ABI_VERSION 5 /* 3: dmi_info grew (pixels_without_depth); dmi_iso_active_cells, DMI_EXCHANGE_PEER_COPY, dmi_multi_peer_chunk
                           * 4: dmi_get_window_pair_count, dmi_get_upload_kernel_ms, dmi_sizeof_info / dmi_sizeof_timings
                           * 5: dmi_get_view_paths */

typedef struct dmi_context dmi_context;

typedef enum dmi_status {
  DMI_OK = 0,
  DMI_ERR_INVALID_ARGUMENT = 1,
  DMI_ERR_DEVICE = 2, /* a HIP runtime call failed; message holds hipGetErrorString */
  DMI_ERR_OUT_OF_MEMORY = 3,
  DMI_ERR_STATE = 4 /* call order violated, e.g. fuse with no views */
} dmi_status;

typedef enum dmi_dtype { DMI_F32 = 0, DMI_F64 = 1 } dmi_dtype;

/* How depth tables are kept in HBM.  AUTO keeps f32 while every uploaded depth is exactly
 * representable in f32 (then f32 storage changes no result bit) and promotes the store to
 * f64 the moment one is not. */
typedef enum dmi_depth_storage { DMI_DEPTH_AUTO = 0, DMI_DEPTH_F32 = 1, DMI_DEPTH_F64 = 2 } dmi_depth_storage;

/* Replaces the grid part of CudaInitialize's arguments (cu:269-272) = the reference's
 * __constant__ c_gridMatrix / c_gridDims / c_gridOrig / c_gridSpacing (cu:55-58). */
typedef struct dmi_grid_desc {
  int32_t cell_dims[3];   /* voxels per axis (vtk point dims - 1) */
  double origin[3];       /* vtkImageData origin (filt.cxx:121-122) */
  double spacing[3];      /* vtkImageData spacing (filt.cxx:125-126) */
  double grid_matrix[16]; /* row-major 4x4, rows = gridVecX/Y/Z (Reconstruction/main.cxx:345-359) */
} dmi_grid_desc;

/* Replaces c_rayPotentialThick/Rho/Eta/Delta (cu:60-63; CudaInitialize cu:273-276). */
typedef struct dmi_ray_potential {
  double thickness;
  double rho;
  double eta;
  double delta;
} dmi_ray_potential;

typedef struct dmi_options {
  int32_t device;         /* HIP device ordinal */
  int32_t grid_dtype;     /* dmi_dtype of the device grid.  DMI_F64 = the reference's contract
                             (ProcessDepthMap<double>, filt.cxx:175); DMI_F32 rounds once per fuse */
  int32_t depth_storage;  /* dmi_depth_storage */
  int32_t count_hits;     /* != 0: keep per-voxel u32 and per-map u64 hit counters (not a reference
                             output; they expose every in-frustum / sentinel decision for parity) */
  int32_t kernel_variant; /* 0 = default; bit field of tuning / test switches (DESIGN.md "kernel_variant"):
                             1 exact division in the general kernel, 2 ignore K structure, 4|8 block shape of
                             the general kernel, 16 never use the tiled kernel, 32..224 tile shape, 256.. the tiled kernel's
                             switches */
  int32_t z_first;        /* this context's grid is the z-slab [z_first, z_first + cell_dims[2]) of a taller grid
                             with the same origin and spacing: voxel k has the centre of global cell z_first + k
                             (cu:78-83 with the global index), so slabs fused on different GPUs are bit-identical
                             to one fusion of the whole grid.  0 = the whole grid */
  void *stream;           /* hipStream_t to run on; NULL = a stream owned by the context */
  void *external_grid;    /* device pointer to a caller-owned grid of grid_dtype[n_voxels]
                             (e.g. a torch tensor that is later all-reduced); NULL = context-owned */
} dmi_options;

typedef struct dmi_timings {
  double last_fuse_kernel_ms; /* hipEvent time of all launches of the last dmi_fuse, on its stream */
  double total_fuse_kernel_ms;
  uint64_t fuse_launches;
  double last_upload_ms; /* host wall time of the last dmi_add_views (copy + convert, synchronised) */
  double last_download_ms; /* host wall time of the last download; for dmi_fuse_range_download the whole call: its slabs' fusions
                              and their copies, which overlap */
  double last_cell_to_point_ms; /* hipEvent time of the last dmi_cell_to_point kernel */
  /* of last_fuse_kernel_ms / total_fuse_kernel_ms, the fusion kernel proper (without the brick classification, the window
   * origins and the workgroup ordering that precede it).  A launch without brick classes -- at most 1024 bricks and fewer than 48
   * views -- is timed as a whole, its one table kernel included: the two are equal then */
  double last_fuse_main_kernel_ms;  /* (a launch without brick classes -- at most 1024 bricks -- is timed as a whole: its one table
                                       kernel included, = last_fuse_kernel_ms) */
  double total_fuse_main_kernel_ms;
} dmi_timings;

typedef struct dmi_info {
  int64_t n_voxels;
  int32_t n_views;
  int32_t depth_width;
  int32_t depth_height;
  int32_t depth_storage_in_use; /* DMI_DEPTH_F32 or DMI_DEPTH_F64 */
  int32_t grid_dtype;
  int32_t k_mode; /* 0 general 4x4 K rows, 1 pinhole with skew, 2 pinhole (chosen from the uploaded Ks) */
  int32_t kernel_variant;
  int32_t tiled_kernel; /* 1: the resident views and the grid meet the preconditions of the register-tiled
                           kernel (axis-aligned grid matrix, pinhole K), 0: the general kernel runs */
  uint64_t device_bytes; /* HBM held by the context */
  uint64_t pixels_without_depth; /* of the resident depth tables, after the best-cost threshold (cu:202's -1; counted on the
                                    device at upload) */
} dmi_info;

/* Fills *opt with the defaults: device 0, f64 grid, AUTO depth storage, no hit counters. */
void dmi_default_options(dmi_options *opt);

/* Replaces CudaInitialize (cu:269-298) and the grid cudaMalloc of ProcessDepthMap (cu:326).
 * The grid starts zero-filled, as RequestData fills it (filt.cxx:133).  opt may be NULL. */
int dmi_create(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_options *opt, dmi_context **out);

/* Replaces the cudaFree/delete block (cu:374-381). */
void dmi_destroy(dmi_context *ctx);

/* Message of the last failing call on ctx; ctx == NULL gives the calling thread's last
 * dmi_create failure.  Never NULL.  Replaces gpuAssert's fprintf+exit (cu:68-76). */
const char *dmi_last_error(const dmi_context *ctx);

/* Replaces the per-depth-map body of ProcessDepthMap (cu:347-360): best-cost threshold
 * (ReconstructionData::ApplyDepthThresholdFilter, RD.cxx:138-167: best_cost > threshold => depth = -1;
 * skipped when best_cost == NULL), marshalling (cu:351-353) and the three H2D copies (cu:358-360).
 * Appends n views; they stay resident in HBM until dmi_clear_views.
 *   depth     [n][H][W] f64 host, vtk point order        ("Depths" array, cu:249)
 *   best_cost [n][H][W] f64 host or NULL                 ("Best Cost Values", RD.cxx:146)
 *   K4, RT4   [n][16]   f64 host, row-major 4x4          (Get4MatrixK / GetMatrixTR, RD.cxx:128-136)
 * All views of a context share W and H (the reference reads them from map 0 only, filt.cxx:167-168). */
int dmi_add_views(dmi_context *ctx, const double *depth, const double *best_cost, double threshold, const double *K4,
                  const double *RT4, int32_t n, int32_t width, int32_t height);

/* Same with f32 depth tables (already thresholded or best_cost given as f32 == NULL only). */
int dmi_add_views_f32(dmi_context *ctx, const float *depth, const double *K4, const double *RT4, int32_t n,
                      int32_t width, int32_t height);

int dmi_clear_views(dmi_context *ctx);

/* Zero the grid (and the hit counters): filt.cxx:133. */
int dmi_reset_grid(dmi_context *ctx);

/* Start from a caller-supplied grid, as ProcessDepthMap uploads io_scalar before accumulating
 * onto it (cu:323-327).  grid: n_voxels f64 host, x fastest. */
int dmi_upload_grid(dmi_context *ctx, const double *grid);

/* Replaces the kernel launches of the depth-map loop (cu:363, one per map in the reference):
 * fuses every resident view into the grid, each voxel accumulated in view order (cu:211).
 * Asynchronous on the context's stream. */
int dmi_fuse(dmi_context *ctx);

/* Fuse only views [first, first+count): lets a caller shard or batch the resident views. */
int dmi_fuse_range(dmi_context *ctx, int32_t first, int32_t count);

/* Fuse every resident view into the cell layers [z_first, z_first + z_count) only.  Lets a caller pipeline a
 * fusion with what consumes the grid (bench.py overlaps the RCCL all-reduce of slab i with the fusion of slab
 * i+1).  Slabs fused one after the other give the same bits as one dmi_fuse.  z_first and z_first + z_count
 * must be multiples of DMI_SLAB_ALIGNMENT or the end of the grid. */
#define DMI_SLAB_ALIGNMENT 32
int dmi_fuse_slab(dmi_context *ctx, int32_t z_first, int32_t z_count);

int dmi_synchronize(dmi_context *ctx);

/* Replace the D2H copy and the per-tuple copy into io_scalar (cu:368-371).  They synchronise.  When the requested type
 * is not the grid's, the conversion runs on the device and the host side is plain copies: `out` in pinned memory
 * (dmi_alloc_pinned, or a buffer the caller registered) is filled at DMA speed, pageable memory at the runtime's
 * staging speed.  The same holds for dmi_upload_grid into an f32 grid. */
int dmi_download_grid_f64(dmi_context *ctx, double *out);
int dmi_download_grid_f32(dmi_context *ctx, float *out);

/* The last step of a chunked reconstruction in one call (cu:343-371: the last maps' kernels, then the copy back): fuse views
 * [first, first + count) -- count may be 0 -- and bring the whole grid to `out` (out_dtype: DMI_F32 or DMI_F64; [n_voxels], pinned
 * memory for DMA speed).  When out_dtype is the grid's own type the grid is fused in n_slabs z-slabs (clamped to 1 .. the number of
 * DMI_SLAB_ALIGNMENT units of the grid) and every slab's copy starts when its fusion ends, on a stream of its own, under the
 * fusion of the next slabs: at 512^3 the copy (9-19 ms) hides all of the fusion but its first slab.  Bit for bit what
 * dmi_fuse_range + dmi_download_grid_* return.  Synchronises.  (Added in round 4; dmi_abi_version() stays 4.) */
int dmi_fuse_range_download(dmi_context *ctx, int32_t first, int32_t count, void *out, int32_t out_dtype, int32_t n_slabs);

/* voxel_hits [n_voxels] u32 and/or map_hits [n_views] u64 (either may be NULL).
 * Needs count_hits at creation. */
int dmi_download_hits(dmi_context *ctx, uint32_t *voxel_hits, uint64_t *map_hits);

/* Device pointer of the grid (context-owned or external) for zero-copy consumers. */
int dmi_grid_device_pointer(dmi_context *ctx, void **ptr);

/* The step right after the filter in the reference's CLI (Reconstruction/main.cxx:151-155): vtkCellDataToPointData
 * over "reconstruction_scalar".  Point (i, j, k) of the (nx+1)(ny+1)(nz+1) lattice gets the mean of its 1..8
 * adjacent cells, accumulated as VTK does (w = 1/count; c += w*v in vtkStructuredData::GetPointCells order), f64.
 * dmi_cell_to_point runs the kernel on the context's stream into a context-owned device buffer (asynchronous);
 * dmi_download_point_data_f64 runs it if the grid changed since, then copies (nx+1)(ny+1)(nz+1) doubles, x fastest,
 * to `out` and synchronises; dmi_point_data_device_pointer hands the device buffer to a zero-copy consumer. */
int dmi_cell_to_point(dmi_context *ctx);
int dmi_download_point_data_f64(dmi_context *ctx, double *out);
int dmi_point_data_device_pointer(dmi_context *ctx, void **ptr);

/* The pre-pass of the step after that (Reconstruction/main.cxx:169-173: vtkContourFilter at `contour` over the point data,
 * i.e. marching cubes over every cell): which cells can produce triangles at all.  A corner is inside when its point value
 * is >= iso (the marching-cubes case bit; a NaN is outside); a cell is ACTIVE when it has both inside and outside corners.
 * *count receives the number of active cells; cell_ids (nullable) the first min(*count, capacity) of them as linear cell
 * ids (k*ny + j)*nx + i in ascending order -- the cells a host marching cubes has to visit, instead of all of them.  Runs
 * dmi_cell_to_point first if the grid changed.  Synchronises. */
int dmi_iso_active_cells(dmi_context *ctx, double iso, uint64_t *count, int64_t *cell_ids, uint64_t capacity);

/* Diagnostic: how many (8 x 8 x column brick, view) pairs of the last dmi_fuse were proven to be handled
 * uniformly.  out[0] mixed (per-voxel path), out[1] all voxels accumulate -eta*rho, out[2] all accumulate 0,
 * out[3] no voxel reaches the accumulate.  All zero when the last fuse ran the general kernel or classes
 * are switched off.  Synchronises. */
int dmi_get_brick_class_histogram(dmi_context *ctx, uint64_t out[4]);

/* Diagnostic: why the mixed pairs of the last dmi_fuse could not be proven uniform.  out[1] non-finite corner value,
 * out[2] the camera plane cuts the brick (c.z <= 0 or too small for the footprint bound), out[3] the footprint is
 * partly outside the depth map, out[4] NaN depths in the footprint, out[5] "no depth" pixels next to depths, out[6]
 * depths within delta of the brick (a surface is near), out[7] every depth of the footprint far behind the brick, with "no depth"
 * pixels among them (free space seen through holes: the FREE column); out[0] unused.  Synchronises. */
int dmi_get_mixed_reason_histogram(dmi_context *ctx, uint64_t out[8]);

/* Diagnostic: how many of the "free space or no depth" pairs (out[7] above) of the last dmi_fuse the fusion kernel served from a
 * window of validity bits (one coalesced fetch per pair) instead of one gather per voxel.  Synchronises. */
int dmi_get_window_pair_count(dmi_context *ctx, uint64_t *out);

/* Which path each resident view takes through the fusion (decided per view from its K, [R|T] and the grid, at dmi_add_views*):
 *   out[0] general kernel (the tiled kernel's bounds do not hold for the view: 6 x slower per view at cfg 3)
 *   out[1] tiled kernel, general K (third row not 0 0 1 0): pixels selected in fp64 only
 *   out[2] tiled kernel, pinhole, fp64 selection only (tier 1 declined: maps beyond 2^24 pixels, bounds not finite)
 *   out[3] tiled kernel, tier 1 with one margin for the view
 *   out[4] tiled kernel, tier 1 with a margin per lane (the camera stands inside or next to the volume)
 *   out[5] of the views counted in [3] and [4]: those with a window record (the window form of the FREE column can serve them)
 * INTEGRATION.md "Magnitudes" says at which coordinate magnitudes a view changes rows. */
int dmi_get_view_paths(dmi_context *ctx, uint64_t out[6]);

/* dmi_get_info / dmi_get_timings fill sizeof(dmi_info) / sizeof(dmi_timings) bytes AS THIS LIBRARY WAS BUILT: a caller compiled
 * against an older header (a shorter struct) must check dmi_abi_version() == DMI_ABI_VERSION -- or compare its own sizeof with
 * dmi_sizeof_info() / dmi_sizeof_timings() -- before passing its buffer. */
int dmi_get_timings(dmi_context *ctx, dmi_timings *out);
int dmi_get_info(dmi_context *ctx, dmi_info *out);
size_t dmi_sizeof_info(void);
size_t dmi_sizeof_timings(void);

/* hipEvent time of the upload pass (the one kernel per staged chunk that thresholds, flips and narrows the tables and builds the
 * pyramid base, the validity bytes and bits, plus the upper pyramid levels) of the last dmi_add_views* call, and summed over
 * the context's life; the copies are not in it.  A call whose tables are staged in several chunks (more than 256 MiB of f64
 * tables) times its LAST chunk and scales it to the call's views: an extrapolation then, a measurement for calls of one chunk
 * (bench.py's calls of 32 views are).  Either pointer may be null. */
int dmi_get_upload_kernel_ms(dmi_context *ctx, double *last, double *total);

/* Pinned host memory for the SoA staging buffers of the host side (hipHostMalloc). */
int dmi_alloc_pinned(size_t bytes, void **out);
int dmi_free_pinned(void *ptr);
/* Diagnostic: the host <-> device copy rates (GB/s, pinned memory, one hipMemcpyAsync of `bytes` each way, best of two)
 * that bound every PCIe-inclusive figure of this path -- the roof next to which bench.py quotes its end-to-end numbers. */
int dmi_pcie_probe(int32_t device, size_t bytes, double *h2d_GBps, double *d2h_GBps);
/* Diagnostic: the fp64 vector rate this device sustains right now (TFLOP/s; a kernel of dependent-free v_fma_f64 chains
 * on every SIMD for about `milliseconds`, best of three).  The fusion kernel is bound by fp64 vector issue, and boxes of
 * the same model differ and drift: bench.py quotes this next to its figures so that runs on different boxes compare. */
int dmi_fp64_probe(int32_t device, double milliseconds, double *tflops);

/* ---- MeshColoration pass (Coloration/MeshColoration.cxx:98-199; the reference runs it on the CPU) ----
 * For every mesh vertex: the views whose projection of the vertex (RD.cxx:169-182: no z-sign test, no depth
 * test) falls inside the image contribute that pixel's RGB (RD.cxx:92-116); outputs are the reference's three
 * point-data arrays "MeanColoration" (u8 x 3, integer mean), "MedianColoration" (u8 x 3) and
 * "NbProjectedDepthMap" (i32), zero where no view sees the vertex.
 *   points [n_points][3] f64 (vtkPoints of the mesh); colors [n_views][H][W][3] u8, the "Color" arrays in vtk
 *   point order; K4, RT4 [n_views][16] row-major (Get4MatrixK / GetMatrixTR).  One-shot: uploads, runs two
 *   kernels, downloads.  Bit-identical to the reference arithmetic (everything after the projection is integer). */
int dmi_color_mesh(const double *points, int64_t n_points, const uint8_t *colors, const double *K4, const double *RT4,
                   int32_t n_views, int32_t width, int32_t height, int32_t device, uint8_t *mean, uint8_t *median,
                   int32_t *count);
const char *dmi_color_last_error(void);

/* The same pass with the colour planes and camera records RESIDENT in HBM: upload the views once, colour any number
 * of vertex sets (BASELINE config 5: the mesh is sharded by vertex across the GPUs, every GPU holds all views, no
 * exchange step).  dmi_color_process works through the vertices in chunks that bound its scratch memory (1 GiB). */
typedef struct dmi_color_context dmi_color_context;
int dmi_color_create(int32_t device, dmi_color_context **out);
void dmi_color_destroy(dmi_color_context *ctx);
/* appends n views: colors [n][H][W][3] u8 in vtk point order, K4 / RT4 [n][16] row-major */
int dmi_color_add_views(dmi_color_context *ctx, const uint8_t *colors, const double *K4, const double *RT4, int32_t n,
                        int32_t width, int32_t height);
int dmi_color_clear_views(dmi_color_context *ctx);
int dmi_color_process(dmi_color_context *ctx, const double *points, int64_t n_points, uint8_t *mean, uint8_t *median,
                      int32_t *count);
/* Upper bound, in bytes, of the device scratch one chunk of vertices may use (default 1 GiB, at least 1024): a smaller
 * budget means more, smaller chunks, never a different result. */
int dmi_color_set_scratch_budget(dmi_color_context *ctx, uint64_t bytes);
/* enable != 0: the vertices of a chunk are worked through along a Z-order curve of their bounding box (device-side key +
 * radix sort), whatever order the caller has them in; inputs and outputs keep the caller's order and no result bit
 * changes.  Worth it for vertices in no particular order (scattered colour gathers become neighbouring ones); a mesh
 * whose vertices already come in a spatially coherent order is faster without.  Default off. */
int dmi_color_set_vertex_reorder(dmi_color_context *ctx, int32_t enable);
/* hipEvent time of the kernels (projection + median) of the last dmi_color_process, summed over its chunks */
int dmi_color_get_kernel_ms(dmi_color_context *ctx, double *out);

/* ---- One fusion over several MI355X of a node (north star: "depth maps shard across the 8 GPUs of one node with a
 * single RCCL all-reduce of the float TSDF grid over xGMI").  The reference has nothing of the kind (one GPU, default
 * stream, cu:302-386); the seam where this plugs in is the pair of driver calls at
 * Reconstruction/vtkCudaReconstructionFilter.cxx:171-176.
 *
 * A dmi_multi_context is `world` ranks, one per GPU, either all inside this process (dmi_multi_create: one thread
 * drives every device, ncclCommInitAll) or one per process (dmi_multi_create_rank: ncclCommInitRank with a unique id
 * that the launcher distributes -- MPI, torch.distributed, a file).  RCCL (librccl.so.1) is loaded on first use;
 * single-GPU users never need it.
 *
 * Partitions:
 *   DMI_PARTITION_VIEWS    rank r fuses its contiguous share of every batch of views into a private full grid, then
 *                          the grids are summed across ranks (exchange below).  Per-voxel summation order changes:
 *                          |result - single-GPU f64 result| <= 2*world*2^-24*sum|partials| for an f32 grid.
 *   DMI_PARTITION_Z_SLABS  rank r owns the cell layers dmi_multi_z_slab(nz, r, world) and fuses ALL views into them:
 *                          no exchange step at all, bit-identical to one single-GPU fusion.
 * Exchange (VIEWS only):
 *   DMI_EXCHANGE_ALL_REDUCE      the contract: every rank ends with the whole summed grid.  The fusion runs in
 *                                n_slabs z-slabs (dmi_fuse_slab) and the all-reduce of slab i runs on a second stream
 *                                while slab i+1 is fused; the last slab is the thinnest (its exchange is the only part
 *                                nothing hides).
 *   DMI_EXCHANGE_REDUCE_SCATTER  for when only the host consumes the grid: rank r ends with the sum of its own 1/world
 *                                of the grid (half the xGMI traffic) and downloads just that.
 *   DMI_EXCHANGE_PEER_COPY       the all-reduce without RCCL and without a compute unit for the transfers (ranks of ONE
 *                                process, dmi_multi_create; at most 16): behind every slab each rank sends the others their
 *                                1/world chunk of it with peer-to-peer copies (the SDMA engines, all xGMI links at once),
 *                                adds what it received to its own chunk IN RANK ORDER with a small kernel queued behind its
 *                                fusion, and copies the sum back into every grid.  Same contract as ALL_REDUCE (every rank
 *                                ends with the whole summed grid) and the same tolerance; unlike a ring's, the order of the
 *                                additions is fixed: the result is the same bits on every run and on every rank.  Several
 *                                of its ranks may share a device (rehearsals on a one-GPU box). */
typedef struct dmi_multi_context dmi_multi_context;

typedef enum dmi_partition { DMI_PARTITION_VIEWS = 0, DMI_PARTITION_Z_SLABS = 1 } dmi_partition;
typedef enum dmi_exchange { DMI_EXCHANGE_ALL_REDUCE = 0, DMI_EXCHANGE_REDUCE_SCATTER = 1, DMI_EXCHANGE_PEER_COPY = 2 } dmi_exchange;

#define DMI_UNIQUE_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct dmi_multi_options {
  int32_t grid_dtype;     /* DMI_F32 (the north star's all-reduce type) or DMI_F64 */
  int32_t depth_storage;  /* dmi_depth_storage */
  int32_t kernel_variant; /* as dmi_options */
  int32_t partition;      /* dmi_partition */
  int32_t exchange;       /* dmi_exchange */
  int32_t n_slabs;        /* z-slabs of the overlapped all-reduce; 0 = default (4), 1 = fuse whole grid, then exchange */
} dmi_multi_options;

typedef struct dmi_multi_info {
  int32_t world;          /* ranks of the fusion = GPUs */
  int32_t n_local;        /* ranks driven by this process */
  int32_t first_rank;     /* rank of local device 0 (the others follow consecutively) */
  int32_t rccl_ranks;     /* what ncclCommCount reports for local rank 0's communicator; 0 = no communicator (Z_SLABS) */
  int32_t rccl_version;   /* ncclGetVersion, 0 when RCCL was never loaded */
  int32_t partition, exchange, n_slabs;
  int64_t n_voxels;       /* of the whole grid */
  int64_t n_views_total;  /* views handed to dmi_multi_add_views so far */
  int64_t n_views_local;  /* of those, resident on this process's devices (VIEWS: the shards; Z_SLABS: all, per device) */
} dmi_multi_info;

typedef struct dmi_multi_timings {
  double last_step_ms;        /* hipEvents on local rank 0's compute stream around one dmi_multi_fuse: reset + fusion +
                                 whatever of the exchange the fusion did not hide */
  double last_fuse_kernel_ms; /* of that, the fusion launches of local rank 0 (dmi_timings.last_fuse_kernel_ms summed
                                 over the slabs) */
  double total_step_ms;
  uint64_t steps;
} dmi_multi_timings;

void dmi_multi_default_options(dmi_multi_options *opt); /* f32 grid, AUTO depth storage, VIEWS + ALL_REDUCE, 4 slabs */

/* Pure partition arithmetic (no GPU needed), the same on every rank:
 * contiguous balanced share [*first, *first + *count) of n items for `rank` of `world` (the first n % world ranks
 * get one more) */
int dmi_multi_view_shard(int64_t n, int32_t rank, int32_t world, int64_t *first, int64_t *count);
/* cell layers [*z_first, *z_first + *z_count) owned by `rank` under DMI_PARTITION_Z_SLABS: boundaries are multiples of
 * DMI_Z_SLAB_ALIGNMENT (16, the tallest voxel column of the fusion kernel: each rank's slab is a grid of its own, created
 * with dmi_options.z_first, not a dmi_fuse_slab range -- those need DMI_SLAB_ALIGNMENT) except the top of the grid; a rank
 * may own nothing when nz is small */
#define DMI_Z_SLAB_ALIGNMENT 16
int dmi_multi_z_slab(int32_t nz, int32_t rank, int32_t world, int32_t *z_first, int32_t *z_count);
/* DMI_EXCHANGE_PEER_COPY: the piece [*first, *first + *count) of an n-element slab that rank `c` sums and hands back
 * (pieces are multiples of 256 elements; the last ranks' may be short or empty) */
int dmi_multi_peer_chunk(int64_t n, int32_t world, int32_t c, int64_t *first, int64_t *count);
/* the z-slabs of the overlapped exchange: writes at most max_slabs (z_first, z_count) pairs, returns how many */
int dmi_multi_slab_ranges(int32_t nz, int32_t n_slabs, int32_t *z_first, int32_t *z_count, int32_t max_slabs);

/* All ranks in this process: devices[0..n) are HIP ordinals, rank i runs on devices[i]. */
int dmi_multi_create(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_multi_options *opt,
                     const int32_t *devices, int32_t n, dmi_multi_context **out);
/* One rank per process.  Rank 0 calls dmi_multi_get_unique_id and the launcher hands the 128 bytes to every rank;
 * all ranks then call dmi_multi_create_rank (collective: returns when every rank has joined). */
int dmi_multi_get_unique_id(uint8_t id[DMI_UNIQUE_ID_BYTES]);
int dmi_multi_create_rank(const dmi_grid_desc *grid, const dmi_ray_potential *ray, const dmi_multi_options *opt,
                          int32_t device, int32_t rank, int32_t world, const uint8_t id[DMI_UNIQUE_ID_BYTES],
                          dmi_multi_context **out);
void dmi_multi_destroy(dmi_multi_context *ctx);
const char *dmi_multi_last_error(const dmi_multi_context *ctx); /* ctx == NULL: the thread's last create failure */

/* Same arguments as dmi_add_views / dmi_add_views_f32, called with the SAME batch on every rank (every process passes
 * the whole batch; each takes what its ranks need: VIEWS -> the rank's share of this batch, Z_SLABS -> all of it). */
int dmi_multi_add_views(dmi_multi_context *ctx, const double *depth, const double *best_cost, double threshold,
                        const double *K4, const double *RT4, int32_t n, int32_t width, int32_t height);
int dmi_multi_add_views_f32(dmi_multi_context *ctx, const float *depth, const double *K4, const double *RT4, int32_t n,
                            int32_t width, int32_t height);
/* Views that belong to local rank `local_index` only, for callers that partition themselves (dmi_multi_view_shard) and
 * never hold the whole batch in one process -- e.g. one process per GPU, each reading its own share of the list files. */
int dmi_multi_add_local_views(dmi_multi_context *ctx, int32_t local_index, const double *depth, const double *best_cost,
                              double threshold, const double *K4, const double *RT4, int32_t n, int32_t width, int32_t height);
int dmi_multi_add_local_views_f32(dmi_multi_context *ctx, int32_t local_index, const float *depth, const double *K4,
                                  const double *RT4, int32_t n, int32_t width, int32_t height);
int dmi_multi_clear_views(dmi_multi_context *ctx);

/* One whole fusion: zero the grids, fuse every resident view, exchange.  Asynchronous; collective across ranks. */
int dmi_multi_fuse(dmi_multi_context *ctx);
int dmi_multi_synchronize(dmi_multi_context *ctx);

/* The fused grid, n_voxels elements, x fastest.  What this process can write, it writes:
 *   ALL_REDUCE: the whole grid (from local rank 0);  REDUCE_SCATTER / Z_SLABS: the parts its ranks own, at their
 *   place in `out` (a single-process context therefore always fills all of `out`).
 * owned_first / owned_count (nullable) receive the contiguous element range this process wrote. */
int dmi_multi_download_grid_f32(dmi_multi_context *ctx, float *out, int64_t *owned_first, int64_t *owned_count);
int dmi_multi_download_grid_f64(dmi_multi_context *ctx, double *out, int64_t *owned_first, int64_t *owned_count);

int dmi_multi_get_info(dmi_multi_context *ctx, dmi_multi_info *out);
int dmi_multi_get_timings(dmi_multi_context *ctx, dmi_multi_timings *out);
/* the single-GPU context of local rank i (views, timings, diagnostics); owned by the multi context */
int dmi_multi_local_context(dmi_multi_context *ctx, int32_t local_index, dmi_context **out);

int dmi_abi_version(void);
int dmi_device_count(void);

#ifdef __cplusplus
}
#endif

#endif /* DMI_H_ */
