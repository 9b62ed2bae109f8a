# make ASAN=1: host code under AddressSanitizer (CPU container only; device code is not instrumented)
CXXFLAGS := --offload-arch=$(ARCH) -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -Wall -Wno-unused-function -Wno-unused-const-variable
LDEXTRA := -fsanitize=address -fno-gpu-sanitize -shared-libsan
