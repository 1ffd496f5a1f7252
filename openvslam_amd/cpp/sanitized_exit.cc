// Linked into the sanitised test programs only (make asan). ROCm's AddressSanitizer runtime carries a device allocator that the HSA runtime's own
// finalizers trip over at process exit (CHECK failed: sanitizer_allocator_device.h "dev_runtime_unloaded_", raised from libhsa-runtime64's static
// destructors, after main returned): the programs therefore leave through _exit() once their own exit handlers have run -- everything the
// sanitizers can say about OUR code has been said by then (leak detection is off for the same reason: the runtimes' allocations are not ours).
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

static void leave(int status, void*) {
    std::fflush(nullptr);
    _exit(status);
}
__attribute__((constructor)) static void register_leave() { on_exit(leave, nullptr); }
