// bvh.h -- drop-in include name of the reference API; every scene-graph class
// is declared in rtk_scene_api.h.
#pragma once
#include "rtk_scene_api.h"
