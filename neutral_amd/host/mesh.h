/*
 * mesh.h -- the Mesh record read at main.c:26-44,62-71,101-106 and
 * neutral_data.c:20-62,109-114.
 */
#ifndef NEUTRAL_AMD_HOST_MESH_H
#define NEUTRAL_AMD_HOST_MESH_H

#include "shared.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  /* problem extent */
  int global_nx;
  int global_ny;
  int local_nx; /* includes 2*pad */
  int local_ny;
  int pad;
  int x_off;
  int y_off;
  double width;
  double height;

  /* time */
  double dt;
  double sim_end;
  int niters;

  /* decomposition (single rank: main.c:42-43) */
  int rank;
  int nranks;
  int ndims;
  int neighbours[NNEIGHBOURS];

  /* geometry; allocated through allocate_data, so device memory when a
   * device kernel set is linked (neutral_data.c:45-62 reads them back with
   * copy_buffer RECV) */
  double* edgex;  /* local_nx + 1 */
  double* edgey;  /* local_ny + 1 */
  double* edgedx; /* local_nx + 1 */
  double* edgedy; /* local_ny + 1 */
  double* celldx; /* local_nx */
  double* celldy; /* local_ny */
} Mesh;

/* uniform rectilinear mesh over [0,width] x [0,height] */
void initialise_mesh_2d(Mesh* mesh);

#ifdef __cplusplus
}
#endif
#endif
