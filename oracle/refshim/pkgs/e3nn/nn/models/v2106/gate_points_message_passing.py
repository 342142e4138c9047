from e3nn import o3


def tp_path_exists(irreps_in1, irreps_in2, ir_out):
    irreps_in1, irreps_in2 = o3.Irreps(irreps_in1).simplify(), o3.Irreps(irreps_in2).simplify()
    ir_out = o3.Irrep(ir_out)
    return any(ir_out in ir1 * ir2 for _, ir1 in irreps_in1 for _, ir2 in irreps_in2)
