/*
 * profiler.h -- wall-clock profiler with the START_PROFILING/STOP_PROFILING
 * surface used at main.c:82,99,115-116 and omp3/neutral.c:575,627.
 */
#ifndef NEUTRAL_AMD_HOST_PROFILER_H
#define NEUTRAL_AMD_HOST_PROFILER_H

#ifdef __cplusplus
extern "C" {
#endif

#define PROFILER_MAX_NAME 128
#define PROFILER_MAX_ENTRIES 256

typedef struct {
  double time;
  int calls;
  char name[PROFILER_MAX_NAME];
} ProfileEntry;

struct Profile {
  double start_seconds;
  int profiler_entry_count;
  ProfileEntry profiler_entries[PROFILER_MAX_ENTRIES];
};

/* global profile used by kernel sets (omp3/neutral.c:575) */
extern struct Profile compute_profile;

void profiler_start_timer(struct Profile* profile);
void profiler_end_timer(struct Profile* profile, const char* entry_name);
void profiler_print_full_profile(struct Profile* profile);

#define START_PROFILING(profile) profiler_start_timer(profile)
#define STOP_PROFILING(profile, name) profiler_end_timer(profile, name)
#define PRINT_PROFILING_RESULTS(profile) profiler_print_full_profile(profile)

#ifdef __cplusplus
}
#endif
#endif
