// persist_util.h - file helpers shared by the two persisted forms (the dense index's `.hipflat` blob, index.hip, and the
// sparse index's blob, sparse_index.hip): the streaming content checksum, whole-buffer read / write, a durable rename.
// Internal to libcqs_hip.so.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <string>

#include <fcntl.h>
#include <libgen.h>
#include <sys/stat.h>
#include <unistd.h>

namespace cqs_persist {

// 64-bit multiply-rotate hash over the 8-byte words of the content, computable in pieces: every piece
// but the last must be a multiple of 8 bytes (the streaming save / load use <= 64 MiB pieces of whole rows).
struct Checksum {
    static constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full;
    uint64_t h;
    explicit Checksum(uint64_t total_bytes) : h(0x27D4EB2F165667C5ull ^ total_bytes) {}
    void update(const void* data, size_t bytes, bool last) {
        const uint8_t* p = (const uint8_t*)data;
        size_t i = 0;
        for (; i + 8 <= bytes; i += 8) {
            uint64_t w;
            memcpy(&w, p + i, 8);
            h ^= w * P1;
            h = ((h << 31) | (h >> 33)) * P2;
        }
        if (last) {
            uint64_t tail = 0;
            if (i < bytes) memcpy(&tail, p + i, bytes - i);
            h ^= tail * P1;
        }
    }
    uint64_t finish() {
        h ^= h >> 29;
        h *= P2;
        h ^= h >> 32;
        return h;
    }
};


inline bool write_all(int fd, const void* data, size_t bytes) {
    const uint8_t* p = (const uint8_t*)data;
    while (bytes) {
        const ssize_t w = write(fd, p, bytes);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; bytes -= (size_t)w;
    }
    return true;
}
inline bool read_all(int fd, void* data, size_t bytes) {
    uint8_t* p = (uint8_t*)data;
    while (bytes) {
        const ssize_t r = read(fd, p, bytes);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        if (r == 0) return false;
        p += r; bytes -= (size_t)r;
    }
    return true;
}
inline void fsync_parent(const std::string& path) {   // make a rename durable (src/cagra.rs:1526-1537)
    std::string tmp = path;
    const char* dir = dirname(&tmp[0]);
    const int fd = open(dir, O_RDONLY | O_DIRECTORY);
    if (fd >= 0) { (void)fsync(fd); close(fd); }
}
inline bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }


}  // namespace cqs_persist
