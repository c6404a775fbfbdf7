// host sanitizer driver: random DAG batches through dvae_batch_create / export / destroy
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dvae_hip.h"
int main() {
    srand(1);
    for (int it = 0; it < 300; ++it) {
        int n = 2 + rand() % 63, T = 2 + rand() % 31, B = 1 + rand() % 40, bn = rand() % 2;
        std::vector<int32_t> types(B * n, 0), ng(B);
        std::vector<uint64_t> mask(B * n, 0);
        for (int b = 0; b < B; ++b) {
            int k = 2 + rand() % (n - 1);
            ng[b] = k;
            types[b * n] = 0;
            for (int v = 1; v < k; ++v) {
                types[b * n + v] = v == k - 1 ? 1 : rand() % T;
                uint64_t m = 1ull << (v - 1);
                for (int u = 0; u + 1 < v; ++u)
                    if (rand() % 4 == 0) m |= 1ull << u;
                mask[b * n + v] = m;
            }
        }
        bool bad = it % 7 == 0;
        if (bad) {  // malformed: must be rejected, never read out of bounds
            int b = rand() % B, which = rand() % 3;
            if (which == 0) types[b * n + 1] = T + 3;
            else if (which == 1) mask[b * n + 1] |= 1ull << (1 + rand() % 5);
            else ng[b] = n + 1;
        }
        void* h = nullptr;
        dvae_batch_counts c;
        int rc = dvae_batch_create(B, n, T, 0, bn, types.data(), mask.data(), ng.data(), &h, &c);
        if (bad) { if (rc == 0) { printf("malformed batch accepted (it=%d)\n", it); dvae_batch_destroy(h); } continue; }
        if (rc != 0) { printf("rejected valid batch: %s\n", dvae_last_error()); return 1; }
        std::vector<int32_t> blob(c.total_ints);
        if (dvae_batch_export(h, blob.data(), c.total_ints) != 0) { printf("export failed\n"); return 1; }
        if (dvae_batch_export(h, blob.data(), c.total_ints - 1) == 0) { printf("short buffer accepted\n"); return 1; }
        dvae_batch_destroy(h);
    }
    printf("sanitizer driver: ok\n");
    return 0;
}
