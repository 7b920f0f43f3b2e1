#!/bin/bash
# GPU box: `build --interval` against the plain build at growing sizes (reads of 150 bp): where do the outputs part?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for n in ${SIZES:-2000000 5000000 8000000}; do
	python - $n <<'PY'
import sys
sys.path.insert(0, ".")
from tools import gen_reads
print(gen_reads.generate(int(sys.argv[1]), "/tmp/iv_reads_%s.txt" % sys.argv[1]))
PY
	./ropebwt3_amd/ropebwt3-amd build -L -d ${EXTRA:-} -o /tmp/iv_a.fmd /tmp/iv_reads_$n.txt 2> /tmp/iv_a.err
	./ropebwt3_amd/ropebwt3-amd build -L -d ${EXTRA:-} --gpus ${NG:-4} --interval -o /tmp/iv_b.fmd /tmp/iv_reads_$n.txt 2> /tmp/iv_b.err
	echo "$n reads: plain $(md5sum /tmp/iv_a.fmd | cut -c1-12) $(stat -c %s /tmp/iv_a.fmd)  interval $(md5sum /tmp/iv_b.fmd | cut -c1-12) $(stat -c %s /tmp/iv_b.fmd)  $(grep -c 'merged the partial' /tmp/iv_b.err) merges; $(grep 'lock-step' /tmp/iv_b.err | cut -c1-150)"
	grep "\[E\|\[W\|ERROR" /tmp/iv_b.err | head -3
	rm -f /tmp/iv_reads_$n.txt
done
