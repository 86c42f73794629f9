/*
 * bdpt_scene.h — host-side scene containers handed to bdpt_set_scene.
 *
 * The reference obtains its scene from Falcor's loader
 * (SharedUtils/SceneLoaderWrapper.cpp:29-108 -> RtScene::loadFromFile).  The only
 * scene it ships is CommonPasses/Data/pink_room/pink_room.fscene and its geometry
 * blob is absent (.MISSING_LARGE_BLOBS), and the BASELINE scenes (Sponza, Bistro,
 * San Miguel) are not in the tree either, so the build supplies seeded procedural
 * scenes through the same container.  CPU-only: no function here touches the GPU.
 */
#ifndef BDPT_SCENE_H_
#define BDPT_SCENE_H_

#include "bdpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bdpt_scene bdpt_scene;

/* Classic 555-unit Cornell box (two blocks, one emissive ceiling patch, one point
 * light below the ceiling), 32 triangles, constant MetalRough materials
 * (SURVEY.md §8d config 1/2). */
bdpt_scene* bdpt_scene_create_cornell(void);

/* Seeded procedural atrium: colonnaded two-storey hall with arches, drapes,
 * relief panels, urns and an alpha-masked lattice; textured (sRGB base colour,
 * roughness/metal, normal map) SpecGloss + MetalRough materials, two point lights
 * and one spot light.  Stand-in for Crytek Sponza at `targetTriangles` (~262k). */
bdpt_scene* bdpt_scene_create_atrium(uint32_t seed, uint32_t targetTriangles);

/* The atrium with `foliageFraction` (0..0.9) of its triangles as alpha-masked, double-sided leaf cards in
 * tree crowns standing in the nave: stand-in for San Miguel's alpha-tested foliage,
 * where most rays run the any-hit alpha test (BDPTUtils.hlsli:115-127) several times. */
bdpt_scene* bdpt_scene_create_courtyard(uint32_t seed, uint32_t targetTriangles, float foliageFraction);

/* The atrium with heavy-tailed triangle areas: floors, slabs and walls are two flat triangles each (up to 165 m^2),
 * the whole budget of `targetTriangles` goes to columns, drapes, urns and ornaments (down to square millimetres) —
 * the same hall and the same count as bdpt_scene_create_atrium, areas spread over six decades as in hand-modelled
 * assets (real Sponza's walls are not tessellated like its drapes).  What an evenly tessellated stand-in hides:
 * large triangles overlapping many small ones in the SAH tree. */
bdpt_scene* bdpt_scene_create_atrium_uneven(uint32_t seed, uint32_t targetTriangles);

/* Uniform random triangle soup in the unit cube (intersection KATs). */
bdpt_scene* bdpt_scene_create_soup(uint32_t seed, uint32_t numTriangles, float maxEdge);

/* Loads a Falcor `.fscene` (JSON: models/instances, lights, cameras) or a bare
 * Wavefront `.obj` (+ `.mtl`; PNG / baseline-JPEG / PPM / PGM / TGA textures), applying the reference's
 * import rules (SceneImporter.cpp:106-460, AssimpModelImporter.cpp:326-417,
 * Material.cpp:119-184, SceneLoaderWrapper.cpp:56-103).  Returns NULL and writes
 * a message into msg (if msgCap > 0) on failure, as loadScene returns nullptr
 * (SceneLoaderWrapper.cpp:60). */
bdpt_scene* bdpt_scene_load(const char* path, char* msg, uint32_t msgCap);

/* Host threads bdpt_scene_load parses and joins a model with (0 = default: BDPT_LOADER_THREADS, else the host's cores,
 * at most 32); returns the previous setting.  The loaded scene does not depend on it, bit for bit
 * (host/SceneLoader.cpp "Host threads of the model loader").  Assimp's importer, which the reference calls
 * (AssimpModelImporter.cpp:509-530), is single-threaded; a 10 M-triangle OBJ is 1.2 GB of text. */
int bdpt_scene_load_threads(int threads);

/* Decode one texture image the way the loader does (PNG, baseline JPEG, PPM/PGM, TGA): RGBA8, row 0 first; *hasAlpha =
 * the file is a 32-bit image for Falcor (Utils/Bitmap.cpp:104-126).  rgba8 may be NULL to query the size.
 * Replaces Bitmap::createFromFile -> FreeImage_Load (Utils/Bitmap.cpp:45-140). */
int bdpt_image_load(const char* path, uint32_t* width, uint32_t* height, uint32_t* hasAlpha, uint8_t* rgba8, uint64_t cap, char* msg,
                    uint32_t msgCap);

/* A Radiance .hdr (RGBE) light probe as RGBA32F, row 0 = top, alpha 1 — what ResourceManager::updateEnvironmentMap
 * (SharedUtils/ResourceManager.cpp:96-110) gets from createTextureFromFile -> FreeImage for the reference's
 * MonValley_*.hdr probes.  Flat and run-length-encoded scanlines, -Y / +Y orientations; mantissa * 2^(e-136) as
 * FreeImage converts.  rgba32f may be NULL to query the size; capFloats counts floats. */
int bdpt_image_load_hdr(const char* path, uint32_t* width, uint32_t* height, float* rgba32f, uint64_t capFloats, char* msg, uint32_t msgCap);

void bdpt_scene_destroy(bdpt_scene* s);

/* Pointers stay valid until bdpt_scene_destroy. */
int bdpt_scene_get_desc(const bdpt_scene* s, bdpt_scene_desc* out);

/* The scene's active camera at the given aspect ratio (SceneLoaderWrapper.cpp:98). */
int bdpt_scene_get_camera(const bdpt_scene* s, float aspect, bdpt_camera* out);

#ifdef __cplusplus
}
#endif
#endif
