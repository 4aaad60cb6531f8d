// vec3.h -- drop-in include name of the reference API; the implementation of
// this vocabulary lives in rtk_math.h.
#pragma once
#include "rtk_math.h"
