"""HBM bytes per C-ABI call from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts 128-byte read requests at 64 bytes,
so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.
Output: a JSON {bench kernel name: bytes per call} (what bench.py's `roofline.traffic` reads) and a CSV with the split.
A "call" is one launch of the family's primary kernel; kernels the same C entry point launches with it (split-K part
adds, the LayerNorm parameter reduction, the attention combine) are charged to that call."""
import collections
import csv
import json
import re
import sys

FAMILIES = [   # (bench name, primary kernel regex, regexes of kernels charged to the same call)
    ('k_gemm16<TN>', r'k_gemm16I.*Lb1ELb1ELi0E', [r'k_add_parts']),
    ('k_gemm16_tn_group', r'k_gemm16_tn_group', [r'k_add_parts_group']),
    ('k_gn_fwd', r'k_gn_apply', [r'k_gn_stats']),
    ('k_gn_bwd', r'k_gn_bwd_dx', [r'k_gn_bwd_sums']),
    ('k_gemm16<NN>', r'k_gemm16I.*Lb0ELb1ELi0E', [r'k_sum_rows']),
    ('k_gemm16<NT>', r'k_gemm16I.*Lb0ELb0ELi0ELi[0-4]E', []),
    ('k_gemm16<NT,conv3x3>', r'k_gemm16I.*Lb0ELb0ELi0ELi5E', []),
    # K20 (fp32 runs): template arguments <A_KS, B_KS, EPI, GATHER> — rocprofv3 prints these kernels demangled (round 6: the
    # mangled-only patterns matched nothing and the fp32 line's roofline.traffic stayed null)
    ('k_gemm32s<NT>', r'k_gemm32sILb0ELb0ELi[012]ELi0E|k_gemm32s<false, false, [012], 0>', []),
    ('k_gemm32s<NT,patch>', r'k_gemm32sILb0ELb0ELi0ELi1E|k_gemm32s<false, false, 0, 1>', []),
    ('k_gemm32s<NT,conv3x3>', r'k_gemm32sILb0ELb0ELi0ELi4E|k_gemm32s<false, false, 0, 4>', []),
    ('k_conv_pad_rows', r'k_(un)?pad_rows', []),
    ('k_gemm32s<NN>', r'k_gemm32sILb0ELb1ELi0ELi0E|k_gemm32s<false, true, 0, 0>', []),
    ('k_gemm32s<NN,dact>', r'k_gemm32sILb0ELb1ELi[34]ELi0E|k_gemm32s<false, true, [34], 0>', []),
    ('k_gemm32s<NN,patch>', r'k_gemm32sILb0ELb1ELi0ELi2E|k_gemm32s<false, true, 0, 2>', []),
    ('k_gemm32s<TN>', r'k_gemm32sILb1ELb1ELi0ELi0E|k_gemm32s<true, true, 0, 0>', [r'k_add_parts32']),
    ('k_gemm32s<TN,patch>', r'k_gemm32sILb1ELb1ELi0ELi3E|k_gemm32s<true, true, 0, 3>', []),
    ('k_gemm32s_tn_group', r'k_gemm32s_tn_group', []),
    ('k_absmax_group', r'k_absmax_group', []),
    ('k_adamw', r'k_adamw', []),
    ('k_sample_select', r'k_sample_select', []),
    ('k_window_attn_bwd', r'k_window_attn(_split)?_bwd', []),
    ('k_window_attn_fwd', r'k_window_attn(_split)?_fwd', []),
    ('k_msda_bwd', r'k_msda_bwd_locattn', [r'k_msda_bwd_value<', r'k_msda_bwd_valueI']),
    ('k_msda_bwd_locattn', r'k_msda_bwd_locattn', []),
    ('k_msda_bwd_value_fx', r'k_msda_bwd_value_fx', [r'k_msda_bwd_relayout']),
    ('k_rowchain', r'k_rowchain', []),
    ('k_msda_fwd_v4', r'k_msda_fwd', []),
    ('k_add_ln_bwd', r'k_add_ln_bwd', [r'k_ln_param_reduce']),
    ('k_add_ln_fwd', r'k_add_ln_fwd', []),
    ('k_attn_bwd', r'k_attn(_split)?_bwd', []),
    ('k_attn_fwd_split', r'k_attn_fwd_split|k_attn_split_fwd', [r'k_attn_fwd_combine', r'k_attn_combine']),
    ('k_wgrad_small', r'k_wgrad_small\(', []),
    ('k_wgrad_small_group', r'k_wgrad_small_group', []),
    ('k_colsum_group', r'k_colsum_group', []),
    ('k_hungarian', r'k_hungarian', []),
    ('k_mask_logits', r'k_mask_logits', []),
    ('k_point_sample_fwd_lds', r'k_point_sample_fwd', []),
    ('k_point_sample_bwd_lds', r'k_point_sample_bwd', []),
    ('k_point_sample_packed', r'k_point_sample_packed', []),
    ('k_mask_loss_rows_fwd', r'k_mask_loss_rows_fwd', []),
    ('k_mask_loss_rows_bwd', r'k_mask_loss_rows_bwd', []),
    ('k_match_cost_terms', r'k_match_terms', []),
    ('k_act_bwd_colsum', r'k_act_bwd_colsum', []),
    ('k_colsum', r'k_colsum<', []),
    ('k_ln_apply', r'k_ln_apply', []),
    ('k_ln_bwd_dense', r'k_ln_bwd_dense', []),
    # round 4: the families bench.py prices beyond this repository's attention / GEMM / loss kernels
    ('k_pfn_stats', r'k_pfn_stats', []),
    ('k_pfn_apply_max', r'k_pfn_apply_max', []),
    ('k_pfn_bwd_route', r'k_pfn_bwd_route', []),
    ('k_pfn_bwd_bn', r'k_pfn_bwd_bn', []),
    ('k_pfn_decorate', r'k_pfn_decorate', []),
    ('k_msda_prepare_fwd', r'k_msda_prepare_fwd', []),
    ('k_msda_prepare_bwd', r'k_msda_prepare_bwd', []),
    ('k_match_products', r'k_match_products', []),
    ('k_skinny_f32', r'k_skinny_f32', []),
    ('k_match_cost', r'k_match_cost', []),
    ('k_copy_group', r'k_copy_group', []),
    ('k_transposed_batch_sum', r'k_transposed_batch_sum', []),
    ('k_upsample_bilinear_bwd', r'k_upsample_bilinear_bwd', []),
    ('k_cls_loss', r'k_cls_loss', []),
    ('k_attn_mask', r'k_attn_mask', []),
    ('k_pack_binary', r'k_pack_binary', []),
    # library kernels by operand type (Tensile names: _BBS_ / _BSS_ / _HHS_ / _HSS_ = 16-bit inputs, _S_B_ / _SB_ = f32)
    ('hipblaslt_16bit', r'^Cijk_.*_(BBS|BSS|HHS|HSS|BS|HS)_', []),
    ('hipblaslt_f32', r'^Cijk_.*_S_B_', []),
    ('miopen_conv', r'^igemm_', [r'batched_transpose', r'SubTensorOp']),
    ('aten_reduce', r'at::native::reduce_kernel|softmax|at::native::.*sort', []),
    ('aten_elementwise', r'at::native::(vectorized_|elementwise_kernel|unrolled_|.*CatArray|.*index_|.*_scatter_gather|.*upsample)', []),
]


# End-of-backward group launches: the graph step flushes them once per captured graph (two calls per step), the eager
# step bench.py instruments once per step — their traffic is recorded PER STEP (listed under "_per_step") and bench.py
# divides by the calls per step it measured.
PER_STEP = ['k_gemm16_tn_group', 'k_gemm32s_tn_group', 'k_wgrad_small_group', 'k_colsum_group']


def load(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out, rows = {}, []
    for name, primary, extra in FAMILIES:
        def total(acc, pats):
            return sum(sum(v) for k, v in acc.items() if any(re.search(p, k) for p in pats))
        calls = sum(len(v) for k, v in fetch.items() if re.search(primary, k))
        if not calls:
            continue
        if name in PER_STEP:
            # = backward passes of the run (one importance-sampling launch each, graph warm-up iterations included)
            calls = sum(len(v) for k, v in fetch.items() if re.search(r'k_sample_select', k)) or calls
        f = 2.0 * total(fetch, [primary] + extra) * 1024 / calls
        w = total(write, [primary] + extra) * 1024 / calls
        out[name] = f + w
        rows.append((name, calls, f / 1e6, w / 1e6, (f + w) / 1e6))
    out['_per_step'] = [n for n in PER_STEP if n in out]
    out['_source'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py '
                      'semantic_kitti_512 B=4 bf16; FETCH_SIZE doubled (gfx950 128-byte requests counted at 64); '
                      'bytes per C-ABI call, averaged over the calls of the run')
    with open(sys.argv[3], 'w') as fh:
        json.dump(out, fh, indent=1)
    with open(sys.argv[4], 'w') as fh:
        fh.write('kernel_family,calls,fetch_MB_per_call(x2 corrected),write_MB_per_call,total_MB_per_call\n')
        for r in rows:
            fh.write(f'"{r[0]}",{r[1]},{r[2]:.2f},{r[3]:.2f},{r[4]:.2f}\n')


if __name__ == '__main__':
    main()
