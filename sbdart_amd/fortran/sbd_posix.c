#define _GNU_SOURCE
/* Process plumbing of the Fortran host's batch mode (sbdart_amd --batch LIST): what ISO_C_BINDING cannot reach in
 * libc without calling variadic functions.  Linked into the executable only -- not part of the engine's C ABI. */
#include <fcntl.h>
#include <stdint.h>
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

/* ---- sbdart_amd --serve: a resident process that owns the GPU and its engines, and the `sbdart` client that stands in
 *      for the reference's executable under RunRT / TestRuns (RunRT.py:2021-2044 Popens `sbdart` in the run's directory
 *      and reads its stdout).  The client hands over its working directory and its OWN file descriptors 1 and 2
 *      (SCM_RIGHTS): the server's run writes straight into them.  Message: int32 length, the directory; reply: int32 exit
 *      code.  The same framing serves the server's private channel to its fork helper (sbd_sv_send_job / recv_job). ---- */
#include <errno.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <sys/un.h>

int sbd_px_dup(int fd) { return dup(fd); }
int sbd_px_dup2(int from, int to) { fflush(stdout); fflush(stderr); return dup2(from, to) < 0 ? -1 : 0; }
int sbd_px_close(int fd) { return close(fd); }
int sbd_px_getpid(void) { return (int)getpid(); }

int sbd_sv_socketpair(int *sv) { return socketpair(AF_UNIX, SOCK_STREAM, 0, sv); }

int sbd_sv_listen(const char *path)
{
    struct sockaddr_un a;
    if (strlen(path) >= sizeof(a.sun_path)) return -1;
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    strcpy(a.sun_path, path);
    /* a socket file nobody listens on is a dead server's: replace it; a live one means another server owns the path */
    const int probe = socket(AF_UNIX, SOCK_STREAM, 0);
    if (probe >= 0) {
        if (connect(probe, (struct sockaddr *)&a, sizeof(a)) == 0) { close(probe); close(fd); errno = EADDRINUSE; return -2; }
        close(probe);
    }
    unlink(path);
    const mode_t old = umask(0177);
    const int rc = bind(fd, (struct sockaddr *)&a, sizeof(a));
    umask(old);
    if (rc < 0 || listen(fd, 64) < 0) { close(fd); return -1; }
    return fd;
}

static int send_all(int fd, const void *p, size_t n)
{
    const char *c = (const char *)p;
    while (n > 0) {
        const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        c += k; n -= (size_t)k;
    }
    return 0;
}
static int recv_all(int fd, void *p, size_t n)
{
    char *c = (char *)p;
    while (n > 0) {
        const ssize_t k = recv(fd, c, n, 0);
        if (k == 0) return 1;                       /* peer closed */
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        c += k; n -= (size_t)k;
    }
    return 0;
}

/* a directory and two file descriptors over a Unix socket */
int sbd_sv_send_job(int sock, const char *dir, int fd1, int fd2)
{
    int32_t len = (int32_t)strlen(dir);
    struct iovec io = { &len, sizeof(len) };
    union { struct cmsghdr h; char buf[CMSG_SPACE(2 * sizeof(int))]; } u;
    struct msghdr m;
    memset(&m, 0, sizeof(m));
    memset(&u, 0, sizeof(u));
    m.msg_iov = &io; m.msg_iovlen = 1;
    m.msg_control = u.buf; m.msg_controllen = sizeof(u.buf);
    struct cmsghdr *c = CMSG_FIRSTHDR(&m);
    c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(2 * sizeof(int));
    int fds[2] = { fd1, fd2 };
    memcpy(CMSG_DATA(c), fds, sizeof(fds));
    for (;;) {
        const ssize_t k = sendmsg(sock, &m, MSG_NOSIGNAL);
        if (k == (ssize_t)sizeof(len)) break;
        if (k < 0 && errno == EINTR) continue;
        return -1;
    }
    return send_all(sock, dir, (size_t)len);
}

/* 0: a job (dir NUL-terminated, *fd1 / *fd2 the sender's descriptors, ours to close); 1: the peer closed; -1: error */
int sbd_sv_recv_job(int sock, char *dir, int dirlen, int *fd1, int *fd2)
{
    int32_t len = 0;
    struct iovec io = { &len, sizeof(len) };
    union { struct cmsghdr h; char buf[CMSG_SPACE(2 * sizeof(int))]; } u;
    struct msghdr m;
    memset(&m, 0, sizeof(m));
    m.msg_iov = &io; m.msg_iovlen = 1;
    m.msg_control = u.buf; m.msg_controllen = sizeof(u.buf);
    ssize_t k;
    do { k = recvmsg(sock, &m, 0); } while (k < 0 && errno == EINTR);
    if (k == 0) return 1;
    if (k != (ssize_t)sizeof(len)) return -1;
    *fd1 = *fd2 = -1;
    for (struct cmsghdr *c = CMSG_FIRSTHDR(&m); c; c = CMSG_NXTHDR(&m, c))
        if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS && c->cmsg_len >= CMSG_LEN(2 * sizeof(int))) {
            int fds[2];
            memcpy(fds, CMSG_DATA(c), sizeof(fds));
            *fd1 = fds[0]; *fd2 = fds[1];
        }
    if (len < 0 || len >= dirlen || *fd1 < 0 || *fd2 < 0) return -1;
    if (recv_all(sock, dir, (size_t)len) != 0) return -1;
    dir[len] = 0;
    return 0;
}

int sbd_sv_send_code(int sock, int code) { int32_t c = code; return send_all(sock, &c, sizeof(c)); }
int sbd_sv_recv_code(int sock, int *code) { int32_t c = 0; const int r = recv_all(sock, &c, sizeof(c)); *code = c; return r; }

/* the next client of the listening socket: its connection (>= 0) with its job, -2 after idle_ms without one, -1 on error */
int sbd_sv_accept(int lfd, int idle_ms, char *dir, int dirlen, int *fd1, int *fd2)
{
    for (;;) {
        struct pollfd p = { lfd, POLLIN, 0 };
        const int r = poll(&p, 1, idle_ms);
        if (r == 0) return -2;
        if (r < 0) { if (errno == EINTR) continue; return -1; }
        const int c = accept(lfd, NULL, NULL);
        if (c < 0) { if (errno == EINTR || errno == ECONNABORTED) continue; return -1; }
        if (sbd_sv_recv_job(c, dir, dirlen, fd1, fd2) == 0) return c;
        close(c);                                   /* a client that said nothing useful: next */
    }
}

/* SBD_DEVICES names the HIP devices of the run ("0", "0,2,3"; "all" = every visible one; unset = "0").  The runtime
 * brings up EVERY agent it can see at its first call -- eight on an 8-GPU node for a run that uses one (VERDICT r05 #3b).
 * Unless the user restricts the runtime himself (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES), the
 * others are hidden here: ROCR_VISIBLE_DEVICES = the list and, since the ordinals then count from 0, SBD_DEVICES = 0..k-1.
 * Must run before the process's first HIP call (the host program's first statement).  Returns 1 when it changed the
 * environment.  SBD_SHOW_DEVICES=1: say on stderr what the runtime will see. */
int sbd_px_restrict_devices(void)
{
    const char *want = getenv("SBD_DEVICES");
    int changed = 0;
    if (!getenv("ROCR_VISIBLE_DEVICES") && !getenv("HIP_VISIBLE_DEVICES") && !getenv("CUDA_VISIBLE_DEVICES")
        && !(want && strcmp(want, "all") == 0)) {
        const char *list = (want && *want) ? want : "0";
        int k = 0, ok = 1;
        for (const char *c = list; *c; ++c) {
            if (*c == ',') ++k;
            else if (*c < '0' || *c > '9') ok = 0;
        }
        if (ok && list[0] != ',' && list[strlen(list) - 1] != ',') {
            char remap[256];
            size_t o = 0;
            for (int i = 0; i <= k && o + 8 < sizeof(remap); ++i) o += (size_t)snprintf(remap + o, sizeof(remap) - o, i ? ",%d" : "%d", i);
            setenv("ROCR_VISIBLE_DEVICES", list, 1);
            setenv("SBD_DEVICES", remap, 1);
            changed = 1;
        }
    }
    if (getenv("SBD_SHOW_DEVICES"))
        fprintf(stderr, "sbdart_amd: devices: ROCR_VISIBLE_DEVICES=%s SBD_DEVICES=%s\n",
                getenv("ROCR_VISIBLE_DEVICES") ? getenv("ROCR_VISIBLE_DEVICES") : "(unset)", getenv("SBD_DEVICES") ? getenv("SBD_DEVICES") : "(unset)");
    return changed;
}
