"""Test seam for pgl_amd.distributed on CPU: an object with the three methods DistGraph's data flow calls on its compute
backend (index / aggregate / gather_rows), written with plain torch CPU ops (index_add_, scatter_reduce).  The product's
only backend is libpglamd (HIP); this one exists so that the partition -> plan -> pack -> exchange -> accumulate data flow,
its transposed (backward) form and the pull/push plans can run under gloo in the CPU container.  It follows the C ABI's
contract for pglamd_aggregate / pglamd_aggregate_ext (include/pgl_amd.h): rows without edges = 0, accumulate modes 0 / 1 / 2,
src/dst scales, the second source table (x2: column ids >= x.shape[0]) and zero_indptr (which rows the zero-fill clears: rows
that are neither written nor cleared are left as NaN here, so a flow that forgets a row fails the comparison)."""
import torch


class TorchBackend(object):
    def index(self, rows, cols, n_rows, edge_ids=None, n_edge_rows=0):
        return (rows.long(), cols.long(), int(n_rows), None if edge_ids is None else edge_ids.long())

    def gather_rows(self, x, idx):
        return x[idx.long()]

    def gather_rows_cast(self, x, idx, dtype, out=None):
        res = (x if idx is None else x[idx.long()]).to(dtype)
        if out is None:
            return res
        out.copy_(res)
        return out

    def row_epilogue(self, z, bias, act, normalize):
        y = z if bias is None else z + bias
        if act == "relu":
            y = torch.relu(y)
        if normalize:
            y = torch.nn.functional.normalize(y, dim=1)
        return y

    def aggregate(self, x, index, reduce_op, n_rows, y=None, message_op="add", src_scale=None, dst_scale=None, out=None,
                   accumulate=0, x2=None, zero_indptr=None):
        rows, cols, _, edge_ids = index
        n_rows = int(n_rows)
        if x2 is not None:
            assert src_scale is None
            x = torch.cat([x, x2], 0)
        if y is not None and edge_ids is not None:
            y = y[edge_ids]
        msg = x[cols]
        tail = tuple(msg.shape[1:])
        if src_scale is not None:
            msg = msg * src_scale[cols].reshape((-1,) + (1,) * len(tail)).to(msg.dtype)
        if y is not None:
            yy = y.reshape((y.shape[0],) + (1,) * (msg.dim() - y.dim()) + tuple(y.shape[1:])) if y.dim() < msg.dim() else y
            msg = {"add": msg + yy, "sub": msg - yy, "mul": msg * yy, "div": msg / yy}[message_op]
            tail = tuple(msg.shape[1:])
        has = torch.zeros(n_rows, dtype=torch.bool)
        has[rows] = True
        if reduce_op in ("sum", "mean"):
            res = torch.zeros((n_rows,) + tail, dtype=msg.dtype).index_add_(0, rows, msg)
            if reduce_op == "mean":
                cnt = torch.bincount(rows, minlength=n_rows).clamp(min=1).to(msg.dtype)
                res = res / cnt.reshape((-1,) + (1,) * len(tail))
        else:
            res = torch.zeros((n_rows,) + tail, dtype=msg.dtype)
            idx = rows.reshape((-1,) + (1,) * len(tail)).expand_as(msg)
            res = res.scatter_reduce(0, idx, msg, "amax" if reduce_op == "max" else "amin", include_self=False)
        if dst_scale is not None:
            res = res * dst_scale.reshape((-1,) + (1,) * len(tail)).to(res.dtype)
        if zero_indptr is not None:
            assert not accumulate and out is None
            empty = zero_indptr[1:] == zero_indptr[:-1]
            keep = has | empty[:n_rows]
            res = torch.where(keep.reshape((-1,) + (1,) * len(tail)), res, torch.full_like(res, float("nan")))
            return res
        if out is None:
            assert not accumulate
            return res
        if accumulate == 0:
            out.copy_(res)
        elif accumulate == 1:
            comb = {"sum": out + res, "mean": out + res, "max": torch.maximum(out, res), "min": torch.minimum(out, res)}[reduce_op]
            out[has] = comb[has]
        else:
            out[has] = res[has]
        return out
