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

/* Uniform random triangle soup in the unit cube (intersection KATs). */
bdpt_scene* bdpt_scene_create_soup(uint32_t seed, uint32_t numTriangles, float maxEdge);

void bdpt_scene_destroy(bdpt_scene* s);

/* Pointers stay valid until bdpt_scene_destroy. */
int bdpt_scene_get_desc(const bdpt_scene* s, bdpt_scene_desc* out);

/* The scene's active camera at the given aspect ratio (SceneLoaderWrapper.cpp:98). */
int bdpt_scene_get_camera(const bdpt_scene* s, float aspect, bdpt_camera* out);

#ifdef __cplusplus
}
#endif
#endif
