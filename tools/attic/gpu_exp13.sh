#!/bin/bash
# loading a large index: peak device memory of `build -i` on the .fmd of 10 M reads (3.02 G symbols)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/gen_reads.py 10000000 /dev/shm/reads10m.txt > /dev/null 2>&1 || python tools/gen_reads.py
ls -la /dev/shm/reads10m.txt | head -2
./ropebwt3_amd/ropebwt3-amd build -L -d -m7g -o /dev/shm/idx.fmd /dev/shm/reads10m.txt 2>&1 | grep -E "Real time|device memory"
ls -la /dev/shm/idx.fmd
head -1000 /dev/shm/reads10m.txt > /dev/shm/few.txt
for c in 16384 1000000; do
RB3GPU_LOAD_CHUNK=$c ./ropebwt3_amd/ropebwt3-amd build -L -d -i /dev/shm/idx.fmd -o /dev/shm/idx2.fmd /dev/shm/few.txt 2>&1 | grep -E "Real time|device memory|in chunks|loaded the index|decoded"
done
