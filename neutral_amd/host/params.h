/*
 * params.h -- deck reader (main.c:29-46, neutral_data.c:24-37,
 * omp3/neutral.c:541).
 *
 * Format (reference problems/NAME.params, problems/neutral.tests):
 *   - `name   value   # comment`      scalar entries
 *   - `name k0=v0 k1=v1 ...`          key/value list entries (source,
 *                                     problem_N, neutral.tests result lines)
 */
#ifndef NEUTRAL_AMD_HOST_PARAMS_H
#define NEUTRAL_AMD_HOST_PARAMS_H

#include "shared.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TERMINATE if the file or the entry is missing */
int get_int_parameter(const char* param_name, const char* filename);
double get_double_parameter(const char* param_name, const char* filename);

/* Finds the line whose first token equals `specifier` and parses its k=v
 * tokens.  keys is a MAX_KEYS x MAX_STR_LEN char matrix.  Returns 1 when the
 * line exists, 0 otherwise (including an unreadable file). */
int get_key_value_parameter(const char* specifier, const char* filename,
                            char* keys, double* values, int* nkeys);

/* non-fatal variants used by the host layer itself */
int try_get_double_parameter(const char* param_name, const char* filename,
                             double* value);

#ifdef __cplusplus
}
#endif
#endif
