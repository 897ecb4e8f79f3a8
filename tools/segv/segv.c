// LD_PRELOAD helper: print a native backtrace on SIGSEGV (no gdb on the GPU boxes).  gcc -shared -fPIC -o libsegv.so segv.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static void handler(int sig, siginfo_t* si, void* ctx) {
  void* bt[64];
  int n = backtrace(bt, 64);
  dprintf(2, "=== signal %d at %p, native backtrace (%d frames):\n", sig, si->si_addr, n);
  backtrace_symbols_fd(bt, n, 2);
  _exit(139);
}
__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  sa.sa_sigaction = handler;
  sigemptyset(&sa.sa_mask);
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  static char stack[1 << 16];
  stack_t ss = {.ss_sp = stack, .ss_size = sizeof(stack), .ss_flags = 0};
  sigaltstack(&ss, NULL);
  sigaction(SIGSEGV, &sa, NULL);
}
