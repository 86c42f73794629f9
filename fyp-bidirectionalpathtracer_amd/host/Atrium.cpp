// placeholder, replaced below
#include "Scene.h"
namespace bdpt { Scene::SharedPtr Scene::createAtrium(uint32_t, uint32_t) { return createCornellBox(); } }
