/*
 * dmi_host.h -- C bindings of the host-side mirror of the reference's operator interface
 * (cudadepthmapintegration_amd/csrc/host/recon_host.h), for callers without a C++ toolchain that matches
 * (tests and bench bind it with ctypes).  Each function names the reference member it stands for:
 *   filt.h / filt.cxx = Reconstruction/vtkCudaReconstructionFilter.{h,cxx}
 *   RD.cxx            = Sources/ReconstructionData.cxx
 *   Helper.h          = Sources/Helper.h
 * The compute entry points proper are in dmi.h; nothing here does TSDF arithmetic.
 */
#ifndef DMI_HOST_H_
#define DMI_HOST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmi_filter dmi_filter; /* a vtkCudaReconstructionFilter (filt.h:48) */

dmi_filter *dmi_filter_new(void);                                    /* vtkCudaReconstructionFilter::New, filt.cxx:74-86 */
void dmi_filter_delete(dmi_filter *f);
void dmi_filter_set_ray_potential_thickness(dmi_filter *f, double v); /* filt.h:57 */
void dmi_filter_set_ray_potential_rho(dmi_filter *f, double v);       /* filt.h:60 */
void dmi_filter_set_ray_potential_eta(dmi_filter *f, double v);       /* filt.h:63 */
void dmi_filter_set_ray_potential_delta(dmi_filter *f, double v);     /* filt.h:66 */
void dmi_filter_set_threshold_best_cost(dmi_filter *f, double v);     /* filt.h:69 */
void dmi_filter_set_file_path_krtd(dmi_filter *f, const char *path);  /* filt.h:73; NULL unsets */
void dmi_filter_set_file_path_vti(dmi_filter *f, const char *path);   /* filt.h:77; NULL unsets */
void dmi_filter_set_grid_matrix(dmi_filter *f, const double m[16]);   /* filt.h:86; row-major vtkMatrix4x4; NULL unsets */
/* SetInputData(vtkImageData*): only dimensions (POINTS), origin and spacing are read (filt.cxx:121-126) */
void dmi_filter_set_input_data(dmi_filter *f, const int32_t dims[3], const double origin[3], const double spacing[3]);
/* In-memory views in place of the two list files: one ReconstructionData each (RD.cxx:55-78):
 * depths / best_cost [H][W] f64 in vtk point order (best_cost may be NULL), K 3x3 and RT 4x4 row-major. */
int dmi_filter_add_view(dmi_filter *f, const double *depths, const double *best_cost, int32_t width, int32_t height,
                        const double K3[9], const double RT[16]);
void dmi_filter_clear_views(dmi_filter *f);
void dmi_filter_set_device(dmi_filter *f, int32_t device);
void dmi_filter_set_kernel_variant(dmi_filter *f, int32_t variant);
/* New and optional (the reference drives one GPU): fuse on several GPUs of the node through dmi_multi_* (dmi.h).
 * n == 0 (the default) = single GPU.  partition: DMI_PARTITION_VIEWS (depth-map shards + one RCCL all-reduce of the
 * f32 grid; the default) or DMI_PARTITION_Z_SLABS (no exchange, f64, bit-identical to one GPU). */
void dmi_filter_set_devices(dmi_filter *f, const int32_t *devices, int32_t n);
void dmi_filter_set_partition(dmi_filter *f, int32_t partition);
/* Pinned host memory of one staging chunk of views (two exist; default 256 MiB): bounds the filter's host memory
 * whatever the number of views -- the list files are read chunk by chunk, as the reference reads them view by view
 * inside its loop (cu:343-353). */
void dmi_filter_set_host_chunk_bytes(dmi_filter *f, uint64_t bytes);
/* != 0: views are read / copied into the staging chunks on the thread that calls Update() (what the VTK binding uses,
 * vtk/vtkCudaReconstructionFilter.cxx: its view source creates VTK readers); 0 (default): on a second thread, while the
 * previous chunk is copied to the device. */
void dmi_filter_set_fill_on_calling_thread(dmi_filter *f, int32_t yes);
/* Update() -> RequestData (filt.cxx:96-151): 1 on success, 0 on error */
int dmi_filter_update(dmi_filter *f);
double dmi_filter_get_execution_time(const dmi_filter *f);            /* filt.h:81 */
double dmi_filter_get_fuse_kernel_ms(const dmi_filter *f);
int64_t dmi_filter_get_number_of_cells(const dmi_filter *f);
/* copies the "reconstruction_scalar" cell array (filt.cxx:129-135) into out[number_of_cells]; returns the count */
int64_t dmi_filter_get_output(const dmi_filter *f, double *out);
const char *dmi_filter_last_error(const dmi_filter *f);

/* Helper.h:105-168.  1 on success, 0 when the file cannot be opened. */
int dmi_read_krtd_file(const char *path, double K3[9], double RT[16]);
/* Helper.h:60-100.  Writes the resolved paths separated by '\n' into buf (NUL-terminated, truncated to
 * buflen) and returns how many entries the list file holds. */
int dmi_extract_all_file_path(const char *list_path, char *buf, size_t buflen);
/* RD.cxx:192-212: 3x3 K into the top-left of a 4x4 identity. */
void dmi_k3_to_k4(const double K3[9], double K4[16]);
/* RD.cxx:138-167 on a bare table; returns how many depths were set to -1. */
int64_t dmi_apply_depth_threshold(double *depths, const double *best_cost, int64_t n, double threshold);
/* RD.cxx:223-229 (vtkXMLImageDataReader) without VTK: every data mode vtkXMLImageDataWriter produces (ascii, binary,
 * appended raw / base64; with or without vtkZLibDataCompressor; UInt32 / UInt64 headers; either byte order), see
 * csrc/host/vti_reader.h.  1 on success.  dims[3]; depths / best_cost sized by the caller (width*height each,
 * best_cost may be NULL); pass depths == NULL to query dims only. */
int dmi_read_depth_map(const char *path, int32_t dims[3], double *depths, double *best_cost, int32_t *has_best_cost);
/* The "Color" array of the same file (RD.cxx:94-95: unsigned char x 3, vtk point order): color sized by the caller
 * (width*height*3) or NULL to query; *has_color = 0 when the file has none.  1 on success. */
int dmi_read_depth_map_color(const char *path, int32_t dims[3], uint8_t *color, int32_t *has_color);

/* MeshColoration(mesh, vtiList, krtdList) + ProcessColoration() (Coloration/MeshColoration.cxx:52-72, :98-199) on mesh
 * points [n_points][3]; fills mean / median [n_points][3] and count [n_points].  1 on success, 0 on error (message in
 * err, truncated to errlen). */
int dmi_mesh_coloration_from_lists(const double *points, int64_t n_points, const char *vti_list, const char *krtd_list,
                                   int32_t device, uint8_t *mean, uint8_t *median, int32_t *count, char *err, size_t errlen);

/* ---- the `Reconstruction` command line (Reconstruction/main.cxx; csrc/host/recon_cli.h) ----
 * What ReadArguments (rmain:216-343) makes of a command line: flags, defaults, validation and the derived grid. */
typedef struct dmi_cli_options {
  int32_t grid_dims[3];       /* POINT dimensions handed to the filter (rmain:118) */
  double grid_spacing[3], grid_origin[3], grid_end[3];
  double grid_matrix[16];     /* CreateGridMatrixFromInput (rmain:345-360), row-major */
  double ray_thick, ray_rho, ray_eta, ray_delta, thresh_best_cost, contour;
  int32_t verbose, summary, force_cubic_voxel;
} dmi_cli_options;
/* 1: the run may proceed, *out filled.  0: an error or --help; the text (what the tool would print) in err. */
int dmi_cli_read_arguments(int32_t argc, const char *const *argv, dmi_cli_options *out, char *err, size_t errlen);
/* The whole tool: ReadArguments, the filter, cell -> point data, meta_image_volume.mha (in the working directory, as the
 * reference), the .vts volume, the summary file.  Process exit code: 0 on success.  No iso-surface (rmain:166-187). */
int dmi_cli_main(int32_t argc, const char *const *argv);

#ifdef __cplusplus
}
#endif

#endif /* DMI_HOST_H_ */
