#define _GNU_SOURCE
/* `sbdart`: the executable RunRT (RunRT.py:29, 2021-2044) and TestRuns/test_runs launch once per run, in the run's
 * directory, reading its stdout -- here a client of a resident `sbdart_amd --serve` process that owns the GPU, its HIP
 * runtime and its engines (a process of its own pays 0.4 s of start-up for a solve of a millisecond).
 *
 *   - connects to the server's Unix socket ($SBDART_AMD_SOCKET, default /tmp/sbdart_amd-<uid>/sock), hands over the
 *     working directory and its OWN file descriptors 1 and 2 (SCM_RIGHTS), waits for the run's exit code;
 *   - no server there: starts one (the sbdart_amd beside this file, detached, idle timeout $SBDART_AMD_IDLE_S, default
 *     300 s) and waits for its socket -- unless SBDART_AMD_NO_AUTOSTART is set;
 *   - no server to be had: execs `sbdart_amd` itself -- one run, one process, the same text.
 * Nothing of the run happens here: no INPUT is read, no number is computed. */
#include <errno.h>
#include <fcntl.h>
#include <libgen.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

int sbd_sv_send_job(int sock, const char *dir, int fd1, int fd2);
int sbd_sv_recv_code(int sock, int *code);

static int try_connect(const char *path)
{
    struct sockaddr_un a;
    if (strlen(path) >= sizeof(a.sun_path)) return -1;
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    strcpy(a.sun_path, path);
    if (connect(fd, (struct sockaddr *)&a, sizeof(a)) == 0) return fd;
    close(fd);
    return -1;
}

int main(int argc, char **argv)
{
    (void)argc;
    char self[PATH_MAX], server[PATH_MAX + 32], sock[PATH_MAX], cwd[PATH_MAX];
    ssize_t n = readlink("/proc/self/exe", self, sizeof(self) - 1);
    if (n <= 0) { strncpy(self, argv[0], sizeof(self) - 1); n = (ssize_t)strlen(self); }
    self[n] = 0;
    snprintf(server, sizeof(server), "%s/sbdart_amd", dirname(self));
    const char *s = getenv("SBDART_AMD_SOCKET");
    if (s && *s) snprintf(sock, sizeof(sock), "%s", s);
    else {
        char dir[64];
        snprintf(dir, sizeof(dir), "/tmp/sbdart_amd-%u", (unsigned)getuid());
        mkdir(dir, 0700);
        snprintf(sock, sizeof(sock), "%s/sock", dir);
    }
    if (!getcwd(cwd, sizeof(cwd))) { perror("sbdart: getcwd"); return 2; }
    int c = try_connect(sock);
    if (c < 0 && !getenv("SBDART_AMD_NO_AUTOSTART")) {
        const pid_t pid = fork();
        if (pid == 0) {                               /* the server: a session of its own, no terminal, no inherited pipes */
            setsid();
            const int nul = open("/dev/null", O_RDWR);
            if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); dup2(nul, 2); if (nul > 2) close(nul); }
            for (int fd = 3; fd < 256; ++fd) close(fd);
            if (chdir("/") != 0) _exit(127);
            execl(server, server, "--serve", sock, (char *)NULL);
            _exit(127);
        }
        for (int i = 0; i < 1500 && c < 0; ++i) {     /* up to 30 s: the server listens only once its runtime is up */
            struct timespec t = { 0, 20 * 1000 * 1000 };
            nanosleep(&t, NULL);
            c = try_connect(sock);
        }
    }
    if (c < 0) {                                      /* no server: this run in a process of its own */
        execl(server, server, (char *)NULL);
        fprintf(stderr, "sbdart: cannot reach a server at %s nor run %s: %s\n", sock, server, strerror(errno));
        return 127;
    }
    fflush(stdout);
    fflush(stderr);
    int code = 0;
    if (sbd_sv_send_job(c, cwd, 1, 2) != 0 || sbd_sv_recv_code(c, &code) != 0) {
        /* the server went away under this run (a STOP inside the model code of a run it served whole): once more, alone */
        close(c);
        execl(server, server, (char *)NULL);
        fprintf(stderr, "sbdart: the server at %s closed the connection and %s cannot be run: %s\n", sock, server, strerror(errno));
        return 127;
    }
    close(c);
    return code;
}
