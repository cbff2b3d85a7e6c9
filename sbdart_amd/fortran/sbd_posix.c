#define _GNU_SOURCE
/* Process plumbing of the Fortran host's batch mode (sbdart_amd --batch LIST): what ISO_C_BINDING cannot reach in
 * libc without calling variadic functions.  Linked into the executable only -- not part of the engine's C ABI. */
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

int sbd_px_chdir(const char *path) { return chdir(path); }

/* file descriptor 1 now writes to `path` (truncated, or appended to); the caller has flushed its Fortran unit 6 */
int sbd_px_stdout_to(const char *path, int append)
{
    fflush(stdout);
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644);
    if (fd < 0) return -1;
    if (dup2(fd, 1) < 0) { close(fd); return -1; }
    close(fd);
    return 0;
}

int sbd_px_stderr_to(const char *path)
{
    fflush(stderr);
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1;
    if (dup2(fd, 2) < 0) { close(fd); return -1; }
    close(fd);
    return 0;
}

int sbd_px_fork(void) { return (int)fork(); }

int sbd_px_wait(int pid)          /* exit code of the child; 128 + signal when it was killed */
{
    int st = 0;
    if (waitpid((pid_t)pid, &st, 0) < 0) return -1;
    return WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
}

void sbd_px_exit_now(int code) { _exit(code); }   /* no atexit handlers, no buffered output written twice */

int sbd_px_exists(const char *path) { return access(path, F_OK) == 0; }

int sbd_px_remove(const char *path) { return unlink(path); }

int sbd_px_touch(const char *path, int value)     /* a marker file holding one integer */
{
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "%d\n", value);
    return fclose(f);
}

int sbd_px_rename(const char *from, const char *to) { return rename(from, to); }

int sbd_px_append_line(const char *path, const char *text)
{
    FILE *f = fopen(path, "a");
    if (!f) return -1;
    fprintf(f, "%s\n", text);
    return fclose(f);
}

/* the caller's file descriptor 1, kept aside while a batch re-points it per run, and put back afterwards */
int sbd_px_stdout_save(void) { fflush(stdout); return dup(1); }
int sbd_px_stdout_restore(int fd)
{
    fflush(stdout);
    if (dup2(fd, 1) < 0) return -1;
    close(fd);
    return 0;
}

/* an environment variable the user has not set: the OpenMP runtime reads its own when it first starts */
#include <stdlib.h>
int sbd_px_setenv_default(const char *name, const char *value) { return setenv(name, value, 0); }

void sbd_px_usleep(int us) { usleep((useconds_t)us); }

int sbd_px_ncpu(void)
{
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) return CPU_COUNT(&set);
    return (int)sysconf(_SC_NPROCESSORS_ONLN);
}

int sbd_px_getcwd(char *buf, int len) { return getcwd(buf, (size_t)len) ? (int)strlen(buf) : -1; }
