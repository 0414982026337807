// worker_gpu.go -- the Go side of libsw_hip.so for the DLSA worker: drop this file into internal/worker/ next to
// worker_impl.go.  It replaces what ExecuteWork (worker_impl.go:108-168) does per pair -- one cgo call per
// (query, target) through findAlignmentWithFallback (:170-202) -- by ONE call per work package
// (include/sw_worker.h: sw_execute_work, or sw_node_execute_work for all GPUs of the machine), and the capacity
// benchmark of cmd/worker/main.go:42-49 by the GPU figure (sw_hip_benchmark_cups / sw_node_benchmark_cups).
// The wire structs, the REST client and the polling loop stay as they are.
//
// NOT COMPILED in the repository that ships it (the build image has no Go toolchain): it is written against the Go
// types of internal/worker/master_rest_client.go:13-68 and the C declarations of include/sw_worker.h, and mirrors
// dlsa_amd/worker.py, which IS tested against the oracle (tests/test_gpu_worker.py).  INTEGRATION.md 6 walks through it.
//
// Link line: point -L / -rpath at the directory that holds libsw_hip.so and -I at this repository's include/.
package worker

// #cgo CFLAGS: -I${SRCDIR}/../../third_party/dlsa_amd/include
// #cgo LDFLAGS: -L${SRCDIR}/../../third_party/dlsa_amd/lib -lsw_hip -Wl,-rpath,${SRCDIR}/../../third_party/dlsa_amd/lib
// #include <stdlib.h>
// #include "sw_worker.h"
import "C"

import (
	"errors"
	"log"
	"unsafe"
)

// GPUNode is the handle of the GPUs this worker process drives (nil: the default device only).
type GPUNode struct {
	h *C.sw_node
}

// NewGPUNode opens every gfx950 device of the machine (sw_node_create(NULL, 0)).
func NewGPUNode() (*GPUNode, error) {
	h := C.sw_node_create(nil, 0)
	if h == nil {
		return nil, errors.New(C.GoString(C.sw_hip_last_error()))
	}
	return &GPUNode{h: h}, nil
}

func (n *GPUNode) Close() {
	if n != nil && n.h != nil {
		C.sw_node_free(n.h)
		n.h = nil
	}
}

// BenchmarkGPU is what the worker registers with the master instead of BenchmarkRust(10 ms, 1, 2)
// (cmd/worker/main.go:43): cells per second of the batch path, integer-valued (the master's field is a pydantic
// int, master/api_models/worker.py:7-16).  node == nil: one GPU.
func BenchmarkGPU(node *GPUNode) (float32, error) {
	var v C.float
	if node != nil {
		v = C.sw_node_benchmark_cups(node.h)
	} else {
		v = C.sw_hip_benchmark_cups()
	}
	if v <= 0 {
		return 0, errors.New(C.GoString(C.sw_hip_last_error()))
	}
	return float32(v), nil
}

// ExecuteWorkGPU computes every pair of `queries` in one library call and posts the results exactly like
// ExecuteWork (worker_impl.go:108-168): buffers of bufferSize, sent when idx%bufferSize == 0 && idx > 0, the rest at
// the end.  Results are bit-identical to the per-pair path (same fallback chain, same int() conversions).
func (w *Worker) ExecuteWorkGPU(work *CompleteWorkPackage, queries []QueryTargetType, node *GPUNode) error {
	w.status = Working
	defer func() { w.status = Waiting }()
	if len(queries) == 0 {
		w.client.SendResult(WorkResult{Alignments: make([]AlignmentDetail, 0)}, *work.ID)
		return nil
	}

	// one C string per distinct sequence of the package; pairs name them by index
	index := make(map[SequenceId]C.uint32_t, len(work.Sequences))
	nSeq := len(work.Sequences)
	if nSeq == 0 { // pairs without sequences: the reference stops the same way (worker_impl.go:119-122)
		log.Fatalf("Target and query not found in work package")
	}
	seqs := (*[1 << 30]*C.char)(C.malloc(C.size_t(nSeq) * C.size_t(unsafe.Sizeof((*C.char)(nil)))))[:nSeq:nSeq]
	i := 0
	defer func() { // (one deferred call, not one per sequence: a package holds 10^5 of them)
		for k := 0; k < i; k++ {
			C.free(unsafe.Pointer(seqs[k]))
		}
		C.free(unsafe.Pointer(&seqs[0]))
	}()
	for id, s := range work.Sequences {
		seqs[i] = C.CString(string(s))
		index[id] = C.uint32_t(i)
		i++
	}
	n := len(queries)
	pairs := (*[1 << 28]C.struct_sw_work_pair)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.struct_sw_work_pair{}))))[:n:n]
	defer C.free(unsafe.Pointer(&pairs[0]))
	for k, comb := range queries {
		q, ok1 := index[comb.Query]
		t, ok2 := index[comb.Target]
		if !ok1 || !ok2 { // worker_impl.go:119-122
			log.Fatalf("Target and query not found in work package")
		}
		pairs[k].query = q
		pairs[k].target = t
	}
	out := (*[1 << 26]C.struct_sw_alignment_detail)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.struct_sw_alignment_detail{}))))[:n:n]
	defer C.free(unsafe.Pointer(&out[0]))

	pkg := C.struct_sw_work_package{
		sequences:        (**C.char)(unsafe.Pointer(&seqs[0])),
		n_sequences:      C.uint64_t(nSeq),
		pairs:            &pairs[0],
		n_pairs:          C.uint64_t(n),
		match_score:      C.int32_t(work.MatchScore),
		mismatch_penalty: C.int32_t(work.MismatchPenalty), // positive on the wire; the library flips the sign (worker_impl.go:126-127)
		gap_penalty:      C.int32_t(work.GapPenalty),
	}
	var rc C.int
	if node != nil {
		rc = C.sw_node_execute_work(node.h, &pkg, &out[0])
	} else {
		rc = C.sw_execute_work(&pkg, &out[0])
	}
	if rc < 0 {
		return errors.New(C.GoString(C.sw_hip_last_error()))
	}
	defer C.sw_alignment_details_free(&out[0], C.uint64_t(n))
	if rc > 0 { // that many pairs failed on all three entry points (entry == -1): the reference dereferences nil there
		log.Printf("%d pairs could not be aligned by any entry point", int(rc))
	}

	bufferSize := 100
	buf := WorkResult{Alignments: make([]AlignmentDetail, 0, bufferSize+1)}
	for idx, comb := range queries {
		d := &out[idx]
		if d.entry < 0 { // no entry point produced a result: nothing is posted for this pair (include/sw_worker.h),
			continue // the master re-schedules what stays missing
		}
		buf.Alignments = append(buf.Alignments, AlignmentDetail{
			Alignment: Alignment{
				QueryAlignment:  C.GoString(d.query_alignment),
				TargetAlignment: C.GoString(d.target_alignment),
				Score:           int(d.score),
				Length:          int(d.length),
				MaxX:            int(d.max_x),
				MaxY:            int(d.max_y),
			},
			TargetQueryCombination: TargetQueryCombination{Query: comb.Query, Target: comb.Target},
		})
		if idx%bufferSize == 0 && idx > 0 {
			go w.client.SendResult(buf, *work.ID)
			buf = WorkResult{Alignments: make([]AlignmentDetail, 0, bufferSize+1)}
		}
	}
	w.client.SendResult(buf, *work.ID)
	return nil
}
