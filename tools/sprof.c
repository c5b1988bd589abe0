/* A sampling profiler for the host side of the element, for images without perf /
 * gdb / valgrind:  gcc -O2 -shared -fPIC tools/sprof.c -o /tmp/sprof.so, then
 *   LD_PRELOAD=/tmp/sprof.so SPROF_OUT=/tmp/sprof.txt gst-launch-1.0 ...
 * and tools/sprof_report.py /tmp/sprof.txt.  ITIMER_PROF ticks (1 kHz of CPU
 * time, process-wide) land on a running thread; the handler keeps the thread id,
 * the interrupted PC and the top 192 words of that thread's stack -- no unwinder
 * in the handler (backtrace() takes the loader's lock and deadlocked against a
 * dlopen in progress).  At exit the samples and /proc/self/maps are written
 * out; the report takes the stack words that point into executable mappings as
 * the probable return addresses and resolves them with nm over the mapped files. */
#define _GNU_SOURCE
#include <ucontext.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <unistd.h>

#define MAX_SAMPLES (1 << 16)
#define DEPTH 192

typedef struct { int tid, n; void *pc; void *stack[DEPTH]; } Sample;
static Sample *samples;
static volatile int n_samples;

static void
on_tick (int sig, siginfo_t *si, void *uc)
{
  (void) sig; (void) si;
  const ucontext_t *u = uc;
  int i = __atomic_fetch_add (&n_samples, 1, __ATOMIC_RELAXED);
  if (i >= MAX_SAMPLES)
    return;
  samples[i].tid = (int) syscall (SYS_gettid);
  samples[i].pc = (void *) u->uc_mcontext.gregs[REG_RIP];
  void **sp = (void **) u->uc_mcontext.gregs[REG_RSP];
  /* (stay within the page run the stack pointer is in: 192 words = 1.5 KB, and
   * a thread's stack is mapped well beyond that above its pointer in practice --
   * the frames of its start routine and the guard are far away) */
  for (int n = 0; n < DEPTH; n++)
    samples[i].stack[n] = sp[n];
  samples[i].n = DEPTH;
}

__attribute__ ((constructor)) static void
sprof_start (void)
{
  if (!getenv ("SPROF_OUT"))
    return;
  samples = calloc (MAX_SAMPLES, sizeof (Sample));
  struct sigaction sa;
  memset (&sa, 0, sizeof sa);
  sa.sa_sigaction = on_tick;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  sigaction (SIGPROF, &sa, NULL);
  const char *hz = getenv ("SPROF_HZ");
  long us = 1000000 / (hz ? atol (hz) : 1000);
  struct itimerval it = { { 0, us }, { 0, us } };
  setitimer (ITIMER_PROF, &it, NULL);
}

__attribute__ ((destructor)) static void
sprof_stop (void)
{
  const char *out = getenv ("SPROF_OUT");
  if (!out || !samples)
    return;
  struct itimerval off = { { 0, 0 }, { 0, 0 } };
  setitimer (ITIMER_PROF, &off, NULL);
  FILE *f = fopen (out, "w");
  if (!f)
    return;
  FILE *m = fopen ("/proc/self/maps", "r");
  char line[1024];
  while (m && fgets (line, sizeof line, m))
    if (strstr (line, " r-xp ") || strstr (line, " r--p "))
      fprintf (f, "M %s", line);
  if (m)
    fclose (m);
  int n = n_samples < MAX_SAMPLES ? n_samples : MAX_SAMPLES;
  for (int i = 0; i < n; i++) {
    fprintf (f, "S %d %p", samples[i].tid, samples[i].pc);
    for (int k = 0; k < samples[i].n; k++)
      fprintf (f, " %p", samples[i].stack[k]);
    fputc ('\n', f);
  }
  fclose (f);
}
