/* oracle/ref_launcher.c -- TEST INFRASTRUCTURE.
 *
 * Runs the unmodified reference: dlopen()s oracle/_ref/libkartref.so (built by
 * oracle/Makefile from the sources under /root/reference/src) with lazy binding
 * and calls the reference's own main() (reference src/main.cpp:87).  Lazy binding
 * is what lets the BAM-only htslib symbols stay unresolved; see oracle/Makefile.
 */
#include <dlfcn.h>
#include <libgen.h>
#include <limits.h>
#include <stdio.h>
#include <unistd.h>

int main(int argc, char **argv)
{
	char exe[PATH_MAX], lib[PATH_MAX + 32];
	ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
	if (n < 0) { perror("readlink"); return 2; }
	exe[n] = '\0';
	snprintf(lib, sizeof(lib), "%s/libkartref.so", dirname(exe));
	void *h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
	if (!h) { fprintf(stderr, "ref_launcher: %s\n", dlerror()); return 2; }
	int (*ref_main)(int, char **) = (int (*)(int, char **))dlsym(h, "main");
	if (!ref_main) { fprintf(stderr, "ref_launcher: no main in %s\n", lib); return 2; }
	return ref_main(argc, argv);
}
