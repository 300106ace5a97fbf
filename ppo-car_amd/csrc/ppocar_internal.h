// ppocar_internal.h -- shared between the translation units of libppocar.so (not installed).
#pragma once
#include <vector>

#include "ppocar.h"

struct pc_track {
    std::vector<double> walls;  // [S][4] x1,y1,x2,y2 pixels, outer segments then inner (car_env.py:653-670)
    std::vector<double> gates;  // [G][4]                      (car_env.py:671-676)
    double start_x = 0, start_y = 0, start_rot = 0;
    int n_walls() const { return (int)(walls.size() / 4); }
    int n_gates() const { return (int)(gates.size() / 4); }
};

int pc_internal_parse_track(const char* path, pc_track* t);
