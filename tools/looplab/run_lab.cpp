// tools/looplab/run_lab.cpp -- loads the code objects tools/looplab/make_lab.py assembled and times their roles.
// usage: run_lab file.hsaco[:label[:waves]] ...   (waves per workgroup of the code object: 8 default, 16)     build: g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include run_lab.cpp -o _run_lab -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
struct Rec { uint64_t cycles; uint32_t hw, role; };
int main(int argc, char** argv) {
    Rec* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(Rec) * 256 * 16) != hipSuccess) return 2;
    const int reps = 2000, grid = 256;
    for (int a = 1; a < argc; a++) {
        std::string path = argv[a], label = path;
        const size_t c = path.find(':');
        int waves = 8;
        if (c != std::string::npos) {
            label = path.substr(c + 1);
            path = path.substr(0, c);
            const size_t c2 = label.find(':');
            if (c2 != std::string::npos) { waves = atoi(label.c_str() + c2 + 1); label = label.substr(0, c2); }
        }
        hipModule_t m;
        hipFunction_t f;
        if (hipModuleLoad(&m, path.c_str()) != hipSuccess || hipModuleGetFunction(&f, m, "_Z3labP3Reciii") != hipSuccess) { printf("%s: cannot load\n", path.c_str()); continue; }
        hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        printf("%-28s", label.c_str());
        for (int mode : {1, 2, 3, 11, 7, 15, 23}) {  // A alone, B alone, both, both with role B at s_setprio 1; +4 = tile-coupled (barrier every 8 trips)
            int md = mode & 7, prio = mode >> 3;
            int r = reps;
            void* args[] = {&d, &r, &md, &prio};
            for (int k = 0; k < 2; k++)
                if (hipModuleLaunchKernel(f, grid, 1, 1, waves * 64, 1, 1, 128 * 1024, nullptr, args, nullptr) != hipSuccess) { printf(" launch failed"); break; }
            hipDeviceSynchronize();
            std::vector<Rec> h(grid * waves);
            hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
            double sum[2] = {0, 0};
            int n[2] = {0, 0};
            double mx = 0;  // the makespan: the slowest wave (waves that are not barrier-coupled finish at different times)
            for (auto& q : h) { sum[q.role & 1] += (double)q.cycles; n[q.role & 1]++; if ((double)q.cycles > mx) mx = (double)q.cycles; }
            const char* nm = mode == 1 ? "A alone" : mode == 2 ? "B alone" : mode == 3 ? "A+B" : mode == 11 ? "A+B prio" : mode == 7 ? "TILES" : mode == 15 ? "TILES prioB" : "TILES prioA";
            if (mode == 1) printf(" | %s %7.1f", nm, sum[0] / n[0] / reps);
            else if (mode == 2) printf(" | %s %7.1f (max %7.1f)", nm, sum[1] / n[1] / reps, mx / reps);
            else if (mode & 4) printf(" | %s %7.1f (max %7.1f)", nm, sum[1] / n[1] / reps, mx / reps);
            else printf(" | %s A %7.1f B %7.1f", nm, sum[0] / n[0] / reps, sum[1] / n[1] / reps);
        }
        printf("   (cycles per trip = per 8 frames)\n");
        hipModuleUnload(m);
    }
    return 0;
}
