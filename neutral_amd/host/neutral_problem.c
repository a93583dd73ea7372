/*
 * neutral_problem.c -- see neutral_problem.h.
 */
#include "neutral_problem.h"

#include "params.h"

#include <string.h>

void neutral_source_from_deck(const char* deck_filename, double mesh_width,
                              double mesh_height, double rank_xpos_0,
                              double rank_ypos_0, double rank_xpos_1,
                              double rank_ypos_1, NeutralSource* source) {
  source->nparticles = get_int_parameter("nparticles", deck_filename);
  source->initial_energy = get_double_parameter("initial_energy", deck_filename);

  char* keys = (char*)malloc(sizeof(char) * MAX_KEYS * MAX_STR_LEN);
  double* values = (double*)malloc(sizeof(double) * MAX_KEYS);
  int nkeys = 0;
  if (!keys || !values ||
      !get_key_value_parameter("source", deck_filename, keys, values, &nkeys) ||
      nkeys < 4) {
    TERMINATE("Parameter file %s did not contain a source entry.\n",
              deck_filename);
  }

  /* the last four values are x, y, width, height as mesh fractions
   * (neutral_data.c:39-43) */
  source->source_xpos = values[nkeys - 4] * mesh_width;
  source->source_ypos = values[nkeys - 3] * mesh_height;
  source->source_width = values[nkeys - 2] * mesh_width;
  source->source_height = values[nkeys - 1] * mesh_height;
  free(keys);
  free(values);

  /* Overlap of the source box with the rank extent.  These exact expressions
   * (neutral_data.c:65-76) enter every injected position, e.g. for the
   * scatter deck they give width 0.6000000000000001, not 0.6, so they are kept
   * term for term. */
  const double sx1 = source->source_xpos + source->source_width;
  const double sy1 = source->source_ypos + source->source_height;
  const double left = max(0.0, source->source_xpos - rank_xpos_0);
  const double bottom = max(0.0, source->source_ypos - rank_ypos_0);
  const double right = max(0.0, rank_xpos_1 - sx1);
  const double top = max(0.0, rank_ypos_1 - sy1);
  source->local_particle_left_off = left;
  source->local_particle_bottom_off = bottom;
  source->local_particle_width =
      max(0.0, (rank_xpos_1 - rank_xpos_0) - (right + left));
  source->local_particle_height =
      max(0.0, (rank_ypos_1 - rank_ypos_0) - (top + bottom));

  /* share of the source inside this rank, rounded to nearest
   * (neutral_data.c:89-95) */
  const double nlocal_real =
      source->nparticles *
      (source->local_particle_width * source->local_particle_height) /
      (source->source_width * source->source_height);
  source->nlocal_particles = (int)(nlocal_real + 0.5);
}

int neutral_cs_file_entries(const char* filename) {
  FILE* fp = fopen(filename, "r");
  if (!fp) {
    return -1;
  }
  int n = 0;
  int ch;
  while ((ch = fgetc(fp)) != EOF) {
    n += (ch == '\n');
  }
  fclose(fp);
  return n;
}

int neutral_read_cs_file(const char* filename, int capacity, double* keys,
                         double* values) {
  FILE* fp = fopen(filename, "r");
  if (!fp) {
    TERMINATE("Could not open the cross section file: %s\n", filename);
  }
  int n = 0;
  while (n < capacity) {
    double k, v;
    if (fscanf(fp, " %lf %lf", &k, &v) != 2) {
      break;
    }
    keys[n] = k;
    values[n] = v;
    n++;
  }
  fclose(fp);
  return n;
}
