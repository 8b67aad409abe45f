// Which kinds of host memory hipHostRegister accepts on this machine: heap, anonymous shared, a /dev/shm file.
// Build: make -C ray-tracer_amd probes (lib/register_probe; run it on the GPU box).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static void attempt(const char* what, void* p, size_t bytes, unsigned flags) {
  hipError_t e = hipHostRegister(p, bytes, flags);
  printf("%-46s flags %u: %s\n", what, flags, hipGetErrorString(e));
  if (e == hipSuccess) {
    void* d = nullptr;
    hipError_t g = hipHostGetDevicePointer(&d, p, 0);
    printf("    device pointer: %s %s\n", hipGetErrorString(g), d == p ? "(same address)" : "(another address)");
    (void)hipHostUnregister(p);
  }
  (void)hipGetLastError();
}
int main() {
  const size_t bytes = (size_t)1920 * 1080 * 3 * 4;
  (void)hipSetDevice(0);
  void* heap = aligned_alloc(4096, bytes);
  memset(heap, 0, bytes);
  attempt("heap", heap, bytes, hipHostRegisterDefault);
  attempt("heap", heap, bytes, hipHostRegisterPortable);
  void* anon = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  attempt("anonymous shared mapping, untouched", anon, bytes, hipHostRegisterDefault);
  memset(anon, 0, bytes);
  attempt("anonymous shared mapping, touched", anon, bytes, hipHostRegisterDefault);
  int fd = open("/dev/shm/rt_amd_register_probe", O_CREAT | O_RDWR, 0600);
  if (fd >= 0 && ftruncate(fd, bytes) == 0) {
    void* shm = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    attempt("/dev/shm file, untouched", shm, bytes, hipHostRegisterDefault);
    memset(shm, 0, bytes);
    attempt("/dev/shm file, touched", shm, bytes, hipHostRegisterDefault);
    attempt("/dev/shm file, touched", shm, bytes, hipHostRegisterPortable);
    attempt("/dev/shm file, touched", shm, bytes, hipHostRegisterMapped);
    void* shm2 = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE | MAP_LOCKED, fd, 0);
    if (shm2 != MAP_FAILED) attempt("/dev/shm file, MAP_POPULATE | MAP_LOCKED", shm2, bytes, hipHostRegisterDefault);
    else printf("MAP_LOCKED mapping failed\n");
    close(fd);
  }
  unlink("/dev/shm/rt_amd_register_probe");
  return 0;
}
