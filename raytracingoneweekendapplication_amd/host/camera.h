// camera.h -- drop-in include name (the reference ships its working camera as
// Camera.txt and includes it as "camera.h", main.cpp:8).
#pragma once
#include "rtk_camera.h"
