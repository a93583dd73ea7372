/*
 * shared_data.h -- per-cell material state shared between arch mini-apps;
 * neutral reads only `density` (main.c:66-70,106).
 */
#ifndef NEUTRAL_AMD_HOST_SHARED_DATA_H
#define NEUTRAL_AMD_HOST_SHARED_DATA_H

#include "shared.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double* density; /* local_nx * local_ny, allocate_data space */
  double* energy;  /* same shape; set from the deck, unused by neutral */
} SharedData;

/* density/energy from the deck's `problem_N density=.. energy=.. xpos=..
 * ypos=.. width=.. height=..` lines, applied in order of N */
void initialise_shared_data_2d(const int local_nx, const int local_ny,
                               const int pad, const double mesh_width,
                               const double mesh_height,
                               const char* problem_def_filename,
                               const double* edgex, const double* edgey,
                               SharedData* shared_data);

#ifdef __cplusplus
}
#endif
#endif
