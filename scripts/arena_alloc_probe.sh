# allocation variants of the level-1 arena, each in processes of its own, the first of them the first process on the box
mkdir -p gpurun_out/r3
{
for rep in 1 2 3; do
  ./build/arena_alloc_probe malloc
  ./build/arena_alloc_probe vmm 1024
  ./build/arena_alloc_probe scrub
done
} > gpurun_out/r3/arena_alloc_probe.txt 2>&1
cat gpurun_out/r3/arena_alloc_probe.txt
