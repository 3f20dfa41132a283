// The ELF reader of the specialiser (rm_jit.h code_object_scratch_bytes: the private_segment_fixed_size of a kernel descriptor, read to decide
// whether a register-capped kernel is kept) under AddressSanitizer + UndefinedBehaviorSanitizer: it parses a file that may come from the
// on-disk kernel cache, i.e. from outside the process.  A synthetic, well-formed ELF64 with one "<kernel>.kd" symbol must give its value;
// the same image truncated at every length, with every byte of its headers flipped, and pure noise must give a number or UINT32_MAX --
// never a read outside the buffer.  CPU only (tests/test_decoder_fuzz_cpu.py builds and runs it).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

struct float4 { float x, y, z, w; };
#include "rm_abi.h"
#include "rm_decode.h"
#include "rm_jit.h"

static void put(std::vector<char>& b, size_t off, const void* p, size_t n) {
    if (b.size() < off + n) b.resize(off + n);
    std::memcpy(b.data() + off, p, n);
}
template <class T> static void put(std::vector<char>& b, size_t off, T v) { put(b, off, &v, sizeof v); }

// sections: 0 null, 1 .rodata (the descriptor, at address 0x1000), 2 .symtab (link 3), 3 .strtab
static std::vector<char> make_elf(uint32_t scratch) {
    std::vector<char> b(64, 0);
    std::memcpy(b.data(), "\177ELF\2\1\1", 7);
    const char names[] = "\0k.kd\0other\0";
    const size_t rodata = 0x100, symtab = 0x180, strtab = 0x200, shoff = 0x240;
    std::vector<char> kd(64, 0);
    std::memcpy(kd.data() + 4, &scratch, 4);
    put(b, rodata, kd.data(), kd.size());
    // symbols: null, "other" (an unrelated one), "k.kd"
    for (int i = 0; i < 3; i++) {
        const size_t s = symtab + 24 * i;
        put<uint32_t>(b, s, i == 1 ? 6u : i == 2 ? 1u : 0u);
        put<uint16_t>(b, s + 6, i ? 1 : 0);
        put<uint64_t>(b, s + 8, i == 2 ? 0x1000u : i == 1 ? 0x1020u : 0u);
        put<uint64_t>(b, s + 16, 64);
    }
    put(b, strtab, names, sizeof names);
    auto sec = [&](int i, uint32_t type, uint64_t addr, uint64_t off, uint64_t size, uint32_t link) {
        const size_t s = shoff + 64 * i;
        put<uint32_t>(b, s + 4, type); put<uint64_t>(b, s + 0x10, addr); put<uint64_t>(b, s + 0x18, off);
        put<uint64_t>(b, s + 0x20, size); put<uint32_t>(b, s + 0x28, link); put<uint64_t>(b, s + 0x38, 0);
    };
    sec(0, 0, 0, 0, 0, 0);
    sec(1, 1, 0x1000, rodata, 64, 0);
    sec(2, 2, 0, symtab, 72, 3);
    sec(3, 3, 0, strtab, sizeof names, 0);
    put<uint64_t>(b, 0x28, shoff);
    put<uint16_t>(b, 0x3A, 64);
    put<uint16_t>(b, 0x3C, 4);
    return b;
}

int main() {
    const std::vector<char> good = make_elf(20u);
    if (rmjit::code_object_scratch_bytes(good, "k") != 20u) { std::printf("well-formed image: wrong value\n"); return 1; }
    if (rmjit::code_object_scratch_bytes(good, "absent") != UINT32_MAX) { std::printf("a kernel that is not there was found\n"); return 1; }
    long parsed = 0, refused = 0;
    auto probe = [&](const std::vector<char>& b) {  // (exact-size copy: ASan sees any read past the end)
        std::vector<char> c(b.begin(), b.end());
        c.shrink_to_fit();
        (rmjit::code_object_scratch_bytes(c, "k") == UINT32_MAX ? refused : parsed)++;
    };
    for (size_t n = 0; n <= good.size(); n++) probe(std::vector<char>(good.begin(), good.begin() + (long)n));  // every truncation
    std::mt19937 rng(99);
    for (size_t i = 0; i < good.size(); i++)  // every byte damaged, three ways
        for (int k = 0; k < 3; k++) {
            std::vector<char> b = good;
            b[i] = k == 0 ? (char)0xFF : k == 1 ? (char)(b[i] ^ (1 << (rng() % 8))) : (char)rng();
            probe(b);
        }
    for (int it = 0; it < 20000; it++) {  // several bytes damaged; noise behind a valid magic
        std::vector<char> b = good;
        if (it % 4 == 3) { b.resize(64 + rng() % 600); for (size_t i = 7; i < b.size(); i++) b[i] = (char)rng(); }
        else for (int k = 0; k < 1 + (int)(rng() % 6); k++) b[rng() % b.size()] = (char)rng();
        probe(b);
    }
    std::printf("elf reader fuzz ok: %ld parsed, %ld refused\n", parsed, refused);
    return parsed > 0 && refused > 0 ? 0 : 1;
}
