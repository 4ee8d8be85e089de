for v in main ldsp; do
  unset MPP_LIB_PATH; [ $v != main ] && export MPP_LIB_PATH=$PWD/mpp_cnn_rs_object_detection_amd/libmppgpu_$v.so
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-convergence --batched-tiles 0 > gpurun_out/cmp_$v.json 2> gpurun_out/cmp_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/cmp_$v.json')); print('$v single', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'matched', d['config']['gt_matched_within_2px'])"
done
