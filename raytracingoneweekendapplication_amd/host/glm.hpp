// glm.hpp -- include name used by the reference (triangle.h:8, mesh.h:6); see glm_min.h.
#pragma once
#include "glm_min.h"
