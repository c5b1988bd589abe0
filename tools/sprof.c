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
#include <dirent.h>
#include <pthread.h>
#include <time.h>

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


static volatile int sampler_stop;

static int
find_threads (const char *name, int *tids, int max)
{
  int n = 0;
  DIR *d = opendir ("/proc/self/task");
  struct dirent *e;
  while (d && (e = readdir (d)) && n < max) {
    char path[320], comm[64] = "";
    if (e->d_name[0] < '0' || e->d_name[0] > '9')
      continue;
    snprintf (path, sizeof path, "/proc/self/task/%s/comm", e->d_name);
    FILE *f = fopen (path, "r");
    if (f) {
      if (fgets (comm, sizeof comm, f) && strstr (comm, name))
        tids[n++] = atoi (e->d_name);
      fclose (f);
    }
  }
  if (d)
    closedir (d);
  return n;
}

static void *
sampler_main (void *arg)
{
  const long us = *(long *) arg;
  const char *name = getenv ("SPROF_THREAD");
  const int pid = getpid ();
  int tids[16], n = 0, since = 1 << 30;
  sigset_t all;
  sigfillset (&all);
  pthread_sigmask (SIG_BLOCK, &all, NULL);
  if (getenv ("SPROF_DELAY_MS")) {      /* (the HIP runtime's start-up does not survive signals) */
    struct timespec d = { atol (getenv ("SPROF_DELAY_MS")) / 1000, atol (getenv ("SPROF_DELAY_MS")) % 1000 * 1000000L };
    nanosleep (&d, NULL);
  }
  while (!sampler_stop) {
    struct timespec ts = { 0, us * 1000 };
    if (since * us > 20000) {           /* (threads come and go) */
      n = find_threads (name, tids, 16);
      since = 0;
    }
    for (int i = 0; i < n; i++)
      syscall (SYS_tgkill, pid, tids[i], SIGPROF);
    nanosleep (&ts, NULL);
    since++;
  }
  return NULL;
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
  if (getenv ("SPROF_THREAD")) {
    /* wall-clock samples of the threads of that name only, sent by a thread of our own
     * (ITIMER_PROF ticks with the scheduler: a few hundred a second, anywhere) */
    static long period_us;
    pthread_t t;
    period_us = us;
    pthread_create (&t, NULL, sampler_main, &period_us);
    pthread_detach (t);
    return;
  }
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
  sampler_stop = 1;
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
