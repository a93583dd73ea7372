/*
 * neutral_problem.h -- problem set-up for the neutral hot path: what the
 * reference's neutral_data.c does between reading the deck and the first
 * solve_transport_2d call, restated as plain functions over host memory so the
 * own driver (neutral_driver.c), bench.py and the tests share one
 * implementation.
 */
#ifndef NEUTRAL_AMD_HOST_NEUTRAL_PROBLEM_H
#define NEUTRAL_AMD_HOST_NEUTRAL_PROBLEM_H

#include "mesh.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  /* deck values (neutral_data.c:24-27) */
  int nparticles;
  double initial_energy;
  /* source box in mesh units (neutral_data.c:39-43) */
  double source_xpos;
  double source_ypos;
  double source_width;
  double source_height;
  /* part of the source box inside this rank (neutral_data.c:65-76) */
  double local_particle_left_off;
  double local_particle_bottom_off;
  double local_particle_width;
  double local_particle_height;
  /* particles this rank injects (neutral_data.c:89-95) */
  int nlocal_particles;
} NeutralSource;

/* Reads nparticles, initial_energy and the `source` entry of the deck and
 * intersects the source box with the rank extent given by the four HOST edge
 * values (x0,y0 = first interior edge, x1,y1 = last).  TERMINATEs when the
 * deck has no source entry (neutral_data.c:33-37). */
void neutral_source_from_deck(const char* deck_filename, double mesh_width,
                              double mesh_height, double rank_xpos_0,
                              double rank_ypos_0, double rank_xpos_1,
                              double rank_ypos_1, NeutralSource* source);

/* Counts the entries of a cross-section file (newline count, as
 * neutral_data.c:129-136).  Returns -1 when the file cannot be opened. */
int neutral_cs_file_entries(const char* filename);

/* Parses up to `capacity` "key value" rows into host arrays; returns the
 * number of rows read (neutral_data.c:149-166 stops early at EOF). */
int neutral_read_cs_file(const char* filename, int capacity, double* keys,
                         double* values);

#ifdef __cplusplus
}
#endif
#endif
