/* rt_mi355x.h - C ABI of librt_mi355x.so, the MI355X (gfx950) path-tracing back end.
 *
 * The reference (FBemf/raytracer @ v0) has no FFI or plugin interface: its hot path is
 * reached through ordinary Rust calls inside one binary crate.  This header is the seam a
 * maintainer would bind (see INTEGRATION.md for the Rust `extern "C"` block): every entry
 * point names the reference code it replaces.  Plain pointers and sizes only; no C++
 * types, no exceptions cross the boundary.  Every `int` result is 0 on success or a
 * negative rt_status; rt_last_error() holds the message.
 *
 * Threading: a context is used by one host thread at a time.  One context drives one GPU
 * (one process per GPU; partial sums are combined by the caller, e.g. with an RCCL
 * reduce over the device buffers that rt_render_accum_device fills).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / scene description fails validation */
    RT_ERR_DEVICE = -2,    /* HIP runtime error (message carries hipGetErrorString) */
    RT_ERR_NO_SCENE = -3,  /* render / trace before rt_upload_scene */
    RT_ERR_SCENE = -4,     /* scene file cannot be read / parsed / constructed */
    RT_ERR_UNSUPPORTED = -5,
    RT_ERR_IO = -6,
    RT_ERR_BUSY = -7       /* rt_render_accum_device_noalloc: another call holds this GPU's static work area */
} rt_status;

/* ------------------------------------------------------------------ flattened scene
 * Replaces the Arc<dyn Hittable> / Arc<dyn Material> / Arc<dyn Texture> object graph that
 * load_config builds (src/config.rs:19-63).  All geometry is f64, as in the reference
 * (src/math.rs:9-13); the library narrows to f32 itself when asked to (RT_PRECISION_F32).
 */

/* BVH node (BVHNode + its leaf Vec, src/hitting.rs:141-229).  Interior: left/right are
 * node indices and lo/hi the box the reference tests (AABB::intersects).  Leaf: left =
 * -1 - payload, right = RT_LEAF_*; lo/hi = bounding_box() of the leaf's object / triangle.  The
 * reference never box-tests a leaf list; the library may use that box, padded, to skip the
 * leaf's primitive test when the ray cannot reach it (conservative: results do not change). */
typedef struct rt_node {
    double lo[3], hi[3];
    int32_t left, right;
} rt_node;
enum { RT_LEAF_OBJECT = 0, RT_LEAF_TRIANGLE = 1 };

/* Rigid transform world -> object space: p' = m * p + b (row-major 3x3).  One record stands
 * for a whole chain of Translate / RotateX / RotateY / RotateZ (src/transforms.rs). */
typedef struct rt_xform {
    double m[9], b[3];
} rt_xform;

typedef struct rt_sphere {        /* src/objects.rs:16-64 */
    double centre[3], radius;
    int32_t material, owner;
} rt_sphere;

typedef struct rt_moving_sphere { /* src/objects.rs:66-135 */
    double centre0[3], centre1[3], time0, time1, radius;
    int32_t material, owner;
} rt_moving_sphere;

typedef struct rt_rect {          /* XYRect / XZRect / YZRect, src/objects.rs:192-382 */
    double a0, a1, b0, b1, k;     /* free-axis ranges (reference order) and the fixed coordinate */
    int32_t axis;                 /* fixed axis: 0 = YZRect, 1 = XZRect, 2 = XYRect */
    int32_t facing_positive;
    int32_t material, owner, sub; /* sub: side index inside a Block (0-5) / Spotlight (0-4) */
    int32_t pad;
} rt_rect;

typedef struct rt_triangle {      /* src/objects.rs:462-565 */
    double point[3], vec1[3], vec2[3], normal[3];
    int32_t material, owner, sub; /* sub: triangle index inside its mesh */
    int32_t pad;
} rt_triangle;

typedef struct rt_plane {         /* src/objects.rs:567-629 */
    double point[3], vec1[3], vec2[3], normal[3], uv_repeat;
    int32_t material, owner;
} rt_plane;

typedef struct rt_medium {        /* ConstantMedium, src/objects.rs:384-460 */
    double neg_inv_density;
    int32_t boundary_object;      /* rt_object index; may be a medium whose own boundary is not one (two levels, src/objects.rs:404-409) */
    int32_t material;             /* phase function */
    int32_t owner;
    uint32_t rng_key;             /* volume-stream key of this medium instance (RNG) */
} rt_medium;

enum {
    RT_OBJ_SPHERE = 0, RT_OBJ_MOVING_SPHERE = 1, RT_OBJ_RECTS = 2, RT_OBJ_TRIANGLE = 3,
    RT_OBJ_PLANE = 4, RT_OBJ_BLAS = 5, RT_OBJ_MEDIUM = 6
};
/* One instance of a scene object: what a BVH leaf / Vec element of the reference holds.
 * RECTS: `count` rectangles from `first`, folded like Vec<Arc<dyn Hittable>>::hit
 * (src/hitting.rs:88-104: same [min,max] for all, strict <, first wins ties) - a Block's six
 * sides, a Spotlight's five panes or one Rect.  BLAS: `first` is the root node of a
 * triangle BVH.  MEDIUM: `first` indexes media.  Others: `first` indexes the array of kind. */
typedef struct rt_object {
    int32_t kind, xform /* -1: none */, first, count;
    int32_t world_index /* position in `world`, -1 for medium boundaries */, pad[3];
} rt_object;

enum {
    RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_DIFFUSE_LIGHT = 3,
    RT_MAT_ISOTROPIC = 4, RT_MAT_CHECKERED = 5
};
typedef struct rt_material {      /* src/materials.rs */
    int32_t type;
    int32_t image;                /* ImageTexture index or -1 for SolidColour (src/textures.rs) */
    double colour[3];             /* solid albedo / emit colour / metal albedo */
    double param;                 /* metal fuzz | dielectric index | checkered tile density */
    int32_t odd, even;            /* checkered parts */
} rt_material;

typedef struct rt_image {         /* ImageTexture, src/textures.rs:27-56: RGB8, top row first */
    int32_t width, height;
    uint64_t texel_offset;        /* byte offset into rt_scene_desc.texels */
} rt_image;

typedef struct rt_camera {        /* Camera, src/camera.rs:11-66 */
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    double lens_radius, start_time, end_time;
} rt_camera;

typedef struct rt_sky {           /* src/config.rs:405-421, src/camera.rs:78-86 */
    int32_t gradient, pad;
    double colour[3];             /* plain colour */
    double unit_dir[3], colour0[3], colour1[3];
} rt_sky;

typedef struct rt_scene_desc {
    const rt_node* nodes;                 uint64_t n_nodes;
    const rt_object* objects;             uint64_t n_objects;
    const rt_xform* xforms;               uint64_t n_xforms;
    const rt_sphere* spheres;             uint64_t n_spheres;
    const rt_moving_sphere* moving_spheres; uint64_t n_moving_spheres;
    const rt_rect* rects;                 uint64_t n_rects;
    const rt_triangle* triangles;         uint64_t n_triangles;
    const rt_plane* planes;               uint64_t n_planes;
    const rt_medium* media;               uint64_t n_media;
    const rt_material* materials;         uint64_t n_materials;
    const rt_image* images;               uint64_t n_images;
    const uint8_t* texels;                uint64_t n_texel_bytes;
    /* Top-level Vec of the world (src/hitting.rs:188-194): entries are object indices, or
     * RT_TOP_BVH for "the BVH rooted at tlas_root".  Folded with strict <, in this order. */
    const int32_t* top_list;              uint64_t n_top;
    int32_t tlas_root;                    /* node index, -1 when the world has no BVH */
    int32_t stack_depth;                  /* traversal stack entries the scene needs */
    rt_camera camera;
    rt_sky sky;
} rt_scene_desc;
enum { RT_TOP_BVH = -1 };

/* ------------------------------------------------------------------ host side: scene files
 * Replaces config::load_config (src/config.rs:19-63): JSON5 scene -> flattened arrays.
 * `asset_root` is prepended to relative texture / mesh file names ("" or NULL = process
 * CWD, the reference behaviour, src/objects.rs:724, src/textures.rs:33). */
typedef struct rt_scene rt_scene;
int rt_scene_load_file(const char* path, const char* asset_root, rt_scene** out);
int rt_scene_load_text(const char* json5_text, const char* asset_root, rt_scene** out);
void rt_scene_free(rt_scene* scene);
const rt_scene_desc* rt_scene_get_desc(const rt_scene* scene);
/* src/main.rs:82: image_height = round(width / aspect). */
uint32_t rt_scene_image_height(const rt_scene* scene, uint32_t width);
/* Error text of the last failed rt_scene_load_* / rt_create on this thread. */
const char* rt_last_global_error(void);

/* ------------------------------------------------------------------ device side */
typedef struct rt_context rt_context;

enum { RT_PRECISION_F64 = 0, RT_PRECISION_F32 = 1 };

typedef struct rt_options {
    int32_t precision;     /* RT_PRECISION_F64 (reference arithmetic, default) or _F32 */
    int32_t lanes;         /* streams (each with its own ray queues) that share the sample batches of a render call, so
                            * that the latency-bound end of one batch runs under the bulk of the next; 0 = default.  The
                            * image does not depend on it bit for bit (sums are added in sample order). */
    uint64_t max_paths;    /* paths in flight per context, shared by the lanes; 0 = default (2^25) */
} rt_options;

int rt_create(int device_id, const rt_options* options /* may be NULL */, rt_context** out);
void rt_destroy(rt_context* ctx);
const char* rt_last_error(const rt_context* ctx);

/* Copies the scene to the context's GPU (the scene is replicated per GPU).  The host
 * arrays may be freed after the call returns. */
int rt_upload_scene(rt_context* ctx, const rt_scene_desc* scene);

/* What the last rt_upload_scene cost ("scene upload / BVH build reported separately"; the
 * reference builds its tree inside load_config, src/config.rs:36 -> src/hitting.rs:147-197, outside its own
 * render timer, src/main.rs:84).  No counterpart in the reference: measurement. */
typedef struct rt_upload_stats {
    double ms_validate;    /* host: every index of the description checked */
    double ms_build;       /* host: traversal layouts derived from `nodes` (two-box nodes, SAH-regrouped four-wide trees, 8-bit quantisation) + their copies */
    double ms_copy;        /* host: record conversion + host-to-device copies of everything else */
    double ms_total;
    uint64_t bytes_device; /* scene bytes resident on the GPU */
    uint32_t n_nodes2, n_nodes4, sah_trees;
    uint32_t needs_uv;     /* some material reads (u, v) (image textures, checkers): hit records carry them (16 B more each way) */
    uint32_t gpu_built_trees; /* mesh trees built by the GPU builder (csrc/device/bvh_build.hip) instead of the host's SAH regrouping */
    uint32_t walk;         /* how rays meet the world: low byte 0 = a tree walk, 1 / 2 / 3 = a sweep of its <= 12 objects through the kernel for rect lists and
                              spheres / for scenes without meshes and media / for every object kind; bits 8..: rects listed one by one for the lane groups */
    double ms_gpu_build;   /* leaf collection on the host + copies + the GPU build of those trees (part of ms_build) */
} rt_upload_stats;
int rt_get_upload_stats(rt_context* ctx, rt_upload_stats* out);

/* Parity entry point.  Replaces `world.hit(ray, tmin, tmax)` (trait Hittable,
 * src/hitting.rs:82-86, as called at src/hitting.rs:21) on explicit rays.  Directions are
 * normalised as Ray::new does (src/math.rs:208-214).  Volume draws use the stream of
 * (seed, pixel = ray index, sample 0, bounce 0). */
typedef struct rt_ray {
    double o[3], d[3], time, tmin, tmax;
} rt_ray;
typedef struct rt_hit {
    int32_t hit;                  /* 0 = miss */
    int32_t owner, sub;           /* primitive identity: scene object index, part index */
    int32_t world_index;          /* `world` entry the hit came through */
    int32_t front_face, material;
    double t, p[3], n[3], u, v;   /* HitRecord, src/hitting.rs:33-41 */
    /* Tie evidence ("Tie handling"): the nearest distance >= tmin at which the ray meets a primitive
     * OTHER than (owner, sub), whatever tmax is; +inf when there is none (and on a miss).  second_t == t is an exact
     * tie: the reference then returns the first element of a list (src/hitting.rs:94) and the later leaf of a BVH
     * (src/hitting.rs:219), which is what (owner, sub) names.  Found by a second, exhaustive walk inside rt_trace. */
    double second_t;
    int32_t inst_id;              /* index into rt_scene_desc.objects of the instance the hit came through; -1 on a miss */
    int32_t pad;
} rt_hit;
int rt_trace(rt_context* ctx, const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out);

/* Camera rays as the row loop builds them (src/main.rs:132-135, Camera::find_ray,
 * src/camera.rs:67-75) for sample `sample` of every pixel of rows [j0, j1), row-major from
 * row j0; j is the reference's bottom-up row index.  tmin = 0.001, tmax = +inf. */
int rt_primary_rays(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1,
                    uint32_t sample, uint64_t seed, rt_ray* out);

typedef struct rt_stats {
    uint64_t rays;        /* closest-hit queries issued by the integrator */
    uint64_t paths;       /* camera paths started */
    double ms_total;      /* GPU time of the call, HIP events on the render stream */
    double ms_extend;     /* time inside the extend (BVH traversal) kernel */
    double ms_shade;      /* time inside the shade (+compaction) kernel */
    uint32_t extend_launches, shade_launches;
    uint32_t batches, max_depth_reached;
    /* Two-kernel bounces only (shade_launches > 0): the launch that carries the last few paths through
     * all their remaining bounces is another kernel (k_bounce) and is kept apart; where one fused kernel
     * runs every bounce it is counted with ms_extend / extend_launches and these stay 0. */
    double ms_tail;       /* time inside tail launches */
    uint64_t rays_tail;   /* closest-hit queries issued inside tail launches (part of `rays`) */
    uint32_t tail_launches;
    uint32_t extend_kernel; /* which kernel ms_extend times: 0 k_extend, 1 k_extend_refill, 2 k_bounce (fused) */
    uint64_t paths_tail;  /* paths that entered tail launches (two-kernel bounces only, like rays_tail) */
    uint64_t paths_first;  /* paths whose camera rays the first bounce launch made itself (sweep scenes): k_raygen neither wrote nor k_bounce read their queue entries */
} rt_stats;
/* ms_extend / ms_shade / ms_tail are filled only while launch timing is on: a HIP event either side of every launch
 * costs the stream about 5 us each (0.7 ms of a 21 ms image), so it is off by default; rays, paths, launch counts
 * and ms_total are always filled.  No counterpart in the reference: measurement. */
int rt_set_launch_timing(rt_context* ctx, int on);

/* Replaces the row closure of src/main.rs:124-137 (+ cast_ray, src/hitting.rs:11-31) for
 * rows [j0, j1) and sample indices [s0, s0 + n_samples) of every pixel: ADDS the radiance
 * sums (not means) into accum, 3 doubles per pixel, row-major from row j0.  accum is DEVICE
 * memory on the context's GPU (e.g. a torch tensor's data_ptr); the caller zeroes it.  The
 * image is identical for any split of the samples over calls / GPUs up to f64 summation
 * order.  Work is enqueued on the context's stream and completed before return. */
int rt_render_accum_device(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1,
                           uint32_t s0, uint32_t n_samples, uint32_t max_bounces, uint64_t seed,
                           double* accum_dev, rt_stats* stats /* may be NULL */);

/* The same call for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc).  It
 * has no counterpart in the reference.  It calls no hipMalloc / hipFree (nor any other allocating routine): the ray queues,
 * hit records and per-path samples live in a static device array of the library (32 MiB per GPU, held from the moment the
 * library is loaded on that GPU until the process ends), sized for RT_NOALLOC_MAX_PATHS = 131072 paths in flight.  Like
 * every render call it follows the live counts from the host and synchronises the context's stream before it returns (so it
 * is not a call to make while a stream is being captured into a graph).  Limit: width * (j1 - j0) <= 131072 pixels per call
 * (RT_ERR_INVALID with a message otherwise); the samples are traced in batches of floor(131072 / pixels) and added in sample
 * order, so accum is bit-identical to rt_render_accum_device's.  Every call on a GPU - from any context - uses that GPU's one
 * array: a call that arrives while another holds it returns RT_ERR_BUSY at once (nothing is traced); make them one after
 * the other.  No statistics, no launch timing. */
#define RT_NOALLOC_MAX_PATHS 131072u
int rt_render_accum_device_noalloc(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1,
                                   uint32_t s0, uint32_t n_samples, uint32_t max_bounces, uint64_t seed,
                                   double* accum_dev);

/* Replaces src/main.rs:138-142 + colour_to_raw (src/main.rs:172-177): mean over
 * spp_total, sqrt, abs / clamp [0, 0.999], floor(255 x).  Both pointers are device memory. */
int rt_quantize_device(rt_context* ctx, const double* accum_dev, uint64_t n_pixels, uint32_t spp_total,
                       uint8_t* rgb8_dev);

/* Host-buffer convenience over the two calls above for one GPU: renders rows [j0, j1) with
 * spp samples; writes (j1-j0)*3*width bytes to rgb8 (row-major from row j0, may be NULL)
 * and the per-pixel radiance sums to sum_rgb (3 doubles per pixel, may be NULL). */
int rt_render_rows(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1,
                   uint32_t spp, uint32_t max_bounces, uint64_t seed, uint8_t* rgb8, double* sum_rgb,
                   rt_stats* stats /* may be NULL */);

/* ------------------------------------------------------------------ several GPUs, one process
 * Replaces the rayon row loop of src/main.rs:117-154 (`into_par_iter` over rows, all cores of one machine) at node
 * scale: the GPUs of device_ids share every band of rows by SAMPLE index (device g traces rt_split_samples(spp, n, g)
 * of every pixel; the scene is replicated), the per-pixel f64 sums are combined on device_ids[0] by one RCCL
 * ncclReduce(sum, f64) per band over xGMI, and that device quantises.  A band's reduce and quantisation run while the
 * next band is traced.  With spp < n the ROWS of a band are split instead and nothing is reduced.  The image equals
 * rt_render_rows' up to the order of the n f64 additions per sum (random streams are keyed by pixel and sample index).
 * One host thread per device is started inside the calls; the rt_multi is used by one caller thread at a time.
 * RCCL is loaded on demand (dlopen); RT_REDUCE=p2p in the environment, or a device list that names a GPU twice (one-GPU
 * rehearsals), sums by peer copies to device_ids[0] and an add kernel instead. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const int* device_ids, int n_devices, const rt_options* options /* may be NULL */, rt_multi** out);
void rt_multi_destroy(rt_multi* m);
int rt_multi_device_count(const rt_multi* m);
const char* rt_multi_last_error(const rt_multi* m /* NULL: the last failed rt_multi_create on this thread */);
int rt_multi_upload_scene(rt_multi* m, const rt_scene_desc* scene);
int rt_multi_set_launch_timing(rt_multi* m, int on);
/* rt_render_rows over all devices.  band_rows: rows per reduce (0 = as many as fit the path budget, i.e. usually the
 * whole call in one band); rgb8 / sum_rgb / stats as in rt_render_rows (stats: sums over the devices, ms_total = the
 * slowest device's). */
int rt_multi_render_rows(rt_multi* m, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t spp,
                         uint32_t max_bounces, uint64_t seed, uint32_t band_rows, uint8_t* rgb8, double* sum_rgb,
                         rt_stats* stats /* may be NULL */);
/* The same with a call-back for finished rows: on_band(user, b0, b1) is called on the caller's thread, in band order, when
 * rows [b0, b1) are final in rgb8 / sum_rgb (the hand-over of `(j, row)` to monitor_progress, src/main.rs:144, 223-238). */
typedef void (*rt_band_callback)(void* user, uint32_t b0, uint32_t b1);
int rt_multi_render_rows_cb(rt_multi* m, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t spp,
                            uint32_t max_bounces, uint64_t seed, uint32_t band_rows, uint8_t* rgb8, double* sum_rgb,
                            rt_stats* stats, rt_band_callback on_band, void* user);
/* Loads RCCL as rt_multi_create would (a copy the process already holds; else the file RT_RCCL_LIB names; else the one next to
 * the HIP runtime this process runs on; else the loader's search path) and returns the path of the file in use, or NULL
 * (rt_multi_last_error(NULL) says why).  Needs no GPU.  For hosts that must keep ONE ROCm stack in the process. */
const char* rt_multi_rccl_library(void);
/* The partition rule: part `part` of `parts` gets `count` consecutive items from `begin`; the ranges cover
 * [first, first + total) exactly once and the first total % parts parts take one extra item. */
void rt_split_samples(uint32_t total, uint32_t parts, uint32_t part, uint32_t first, uint32_t* begin, uint32_t* count);

/* ------------------------------------------------------------------ host I/O (no GPU)
 * The file formats either side of the row loop, for hosts that keep the reference's
 * command-line behaviour (bin/raytracer does).  Errors: rt_last_global_error(). */

/* PartFile::new (src/part_file.rs:13-30) + the "<W> <H>\n" header monitor_progress writes
 * (src/main.rs:197-200): creates "<basename(image_name)>.<n>.PART" in the working directory
 * with the first free n. */
typedef struct rt_part_file rt_part_file;
int rt_part_file_create(const char* image_name, uint32_t width, uint32_t height, rt_part_file** out);
const char* rt_part_file_path(const rt_part_file* part);
/* PartFile::write_part (src/part_file.rs:32-39): "L<j> " + row + "\n", flushed.  line_number is
 * the bottom-up row index j of src/main.rs:121. */
int rt_part_file_write(rt_part_file* part, uint32_t line_number, const uint8_t* row, size_t row_bytes);
/* Closes the file and frees the handle; remove != 0 deletes it (src/main.rs:239-241). */
int rt_part_file_finish(rt_part_file* part, int remove);
/* PartFile::read (src/part_file.rs:41-99).  rows: image_height * image_width * 3 bytes indexed
 * by j; present[j] = 1 for rows found in the file.  Fails with the reference's messages
 * (dimension mismatch, "Missing leading L", ...) unless recover_corrupt stops at the first bad
 * record instead. */
int rt_part_file_read(const char* name, uint32_t image_height, uint32_t image_width, int recover_corrupt,
                      uint8_t* rows, uint8_t* present);

/* img.save(output_file) (src/main.rs:161-162): rgb8 is width*height*3 bytes, top row first; the encoder follows the
 * extension: .png, .jpg / .jpeg (baseline JFIF, quality 75), .ppm / .pnm, .pam, .bmp, .tga, .tif / .tiff (one uncompressed
 * strip), .ff (farbfeld); anything else is RT_ERR_IO ("not supported"). */
int rt_save_image(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8);

/* TimedProgressBar (src/progress.rs:12-115) drawing on a duplicate of file descriptor fd. */
typedef struct rt_progress rt_progress;
int rt_progress_create(int fd, size_t length, const char* label, const char* glyphs, uint32_t number_samples,
                       uint32_t times_will_be_updated, rt_progress** out);
void rt_progress_update(rt_progress* progress);
void rt_progress_clear(rt_progress* progress);
void rt_progress_free(rt_progress* progress);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
