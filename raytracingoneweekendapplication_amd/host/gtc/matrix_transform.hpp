// gtc/matrix_transform.hpp -- include name used by the reference (triangle.h:9, mesh.h:7); see glm_min.h.
#pragma once
#include "../glm_min.h"
