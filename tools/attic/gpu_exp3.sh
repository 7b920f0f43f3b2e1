#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for U in tools/ubench/lfchase tools/ubench/lfchase60; do
echo "=== $U"
for st in 0 1; do
for mode in 0 1 2; do
	for w in 22912 45824 65536; do
		$U 90 162000 $mode $w 400 $st
	done
done
done
echo "--- small index (L2-resident: 4 MB slots, 1k groups)"
for mode in 0 1 2; do $U 4 1100 $mode 22912 400 1; done
echo "--- huge index (545 MB slots)"
for mode in 0 1 2; do $U 545 1000000 $mode 65536 400 1; done
done
