/* LD_PRELOAD helper: native backtrace on SIGSEGV (debugging aid for sporadic host-side crashes). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig, siginfo_t *si, void *ctx) {
    void *frames[64];
    char msg[128];
    int n = snprintf(msg, sizeof msg, "\n*** signal %d at address %p, native backtrace:\n", sig, si ? si->si_addr : 0);
    (void)!write(2, msg, (size_t)n);
    n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    void *dummy[4];
    backtrace(dummy, 4); /* pre-load libgcc */
}
