#!/bin/bash
# READ-ONLY probe of what the leased box says about compute / memory partitions of its MI355X. Changes nothing.
export TMPDIR=/tmp
out=gpurun_out/r06_partition_probe.txt
mkdir -p gpurun_out
{
echo "== id / caps"; id; grep -i cap /proc/self/status
echo "== devices"; ls -la /dev/kfd /dev/dri 2>&1
echo "== sysfs partition files"
for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition \
         /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/available_memory_partition; do
  [ -e "$f" ] && { echo "$f: $(cat $f 2>&1)  [$(stat -c '%A %U' $f)]  writable=$([ -w $f ] && echo yes || echo no)"; }
done
echo "== mount of /sys"; grep -E " /sys( |/)" /proc/mounts | head -5
echo "== rocm-smi"; timeout 60 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -30
echo "== amd-smi partition"; timeout 60 amd-smi partition --current 2>&1 | head -40
timeout 60 amd-smi partition --accelerator 2>&1 | head -60
echo "== amd-smi list"; timeout 60 amd-smi list 2>&1 | head -30
echo "== rocminfo agents"; timeout 60 rocminfo 2>&1 | grep -E "Marketing Name|Compute Unit|Name: +gfx|Uuid" | head -40
echo "== kfd topology nodes"; ls /sys/class/kfd/kfd/topology/nodes/ 2>&1
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n: $(grep -E 'simd_count|num_xcc|cu_count' $n/properties 2>/dev/null | tr '\n' ' ')"; done
echo "== torch"; python -c "import torch;print(torch.cuda.device_count(), [torch.cuda.get_device_properties(i).multi_processor_count for i in range(torch.cuda.device_count())])"
echo "== rccl env"; env | grep -iE "nccl|rccl|hsa|hip|rocr" 
} > $out 2>&1
cat $out
