"""Test seam for pgl_amd.distributed on CPU: an object with the three methods DistGraph's data flow calls on its compute
backend (index / aggregate / gather_rows), written with plain torch CPU ops (index_add_, scatter_reduce).  The product's
only backend is libpglamd (HIP); this one exists so that the partition -> plan -> pack -> exchange -> accumulate data flow,
its transposed (backward) form and the pull/push plans can run under gloo in the CPU container.  It follows the C ABI's
contract for pglamd_aggregate / pglamd_aggregate_ext (include/pgl_amd.h): rows without edges = 0, accumulate modes 0 / 1 / 2,
src/dst scales, the second source table (x2: column ids >= x.shape[0]) and zero_indptr (which rows the zero-fill clears: rows
that are neither written nor cleared are left as NaN here, so a flow that forgets a row fails the comparison), and for the wire
mirror of pglamd_aggregate_wire (the rows a launch stores also land in the next aggregation's send buffer; the buffers start as
NaN in the tests, so a slot nobody wrote fails the comparison too)."""
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

    @staticmethod
    def _mirror(wire, written, values):
        """The wire contract of pglamd_aggregate_wire / pglamd_row_epilogue_wire (include/pgl_amd.h): every row the launch STORED
        (`written`) goes -- times wire.scale -- to its slots of the send buffer (two buffers when the descriptor is split) and to
        the dense scaled copy.  Slots of rows that were not stored keep their contents."""
        n_rows = int(values.shape[0])
        desc, more = wire.desc.long(), wire.more.long()
        rows, poss = [], []
        for r in torch.nonzero(written & (desc[:, 0] > 0)).reshape(-1).tolist():   # {count, p0, p1, p2}: include/pgl_amd.h
            c = int(desc[r, 0])
            ps = desc[r, 1:1 + c].tolist() if c <= 3 else desc[r, 1:3].tolist() + more[int(desc[r, 3]):int(desc[r, 3]) + c - 2].tolist()
            rows += [r] * c
            poss += ps
        r = torch.tensor(rows, dtype=torch.long)
        pos = torch.tensor(poss, dtype=torch.long)
        v = values[r]
        if wire.scale is not None:
            v = v * wire.scale[r].reshape(-1, 1).to(v.dtype)
        if wire.split:
            wire.buf[pos] = v[:, :wire.split]
            wire.buf2[pos] = v[:, wire.split:]
        else:
            wire.buf[pos] = v
        if wire.scaled_out is not None:
            wire.scaled_out[written] = values[written] * wire.scale[written].reshape(-1, 1).to(values.dtype)

    def row_epilogue(self, z, bias, act, normalize, wire=None):
        y = z if bias is None else z + bias
        if act == "relu":
            y = torch.relu(y)
        if normalize:
            y = torch.nn.functional.normalize(y, dim=1)
        if wire is not None:
            self._mirror(wire, torch.ones(y.shape[0], dtype=torch.bool), y.detach())
        return y

    def aggregate(self, x, index, reduce_op, n_rows, y=None, message_op="add", src_scale=None, dst_scale=None, out=None,
                  accumulate=0, x2=None, zero_indptr=None, wire=None):
        if wire is None:
            return self._aggregate(x, index, reduce_op, n_rows, y, message_op, src_scale, dst_scale, out, accumulate, x2, zero_indptr)
        assert y is None and src_scale is None and reduce_op in ("sum", "mean")
        rows = index[0]
        n_rows = int(n_rows)
        has = torch.zeros(n_rows, dtype=torch.bool)
        has[rows] = True
        res = self._aggregate(x, index, reduce_op, n_rows, y, message_op, src_scale, dst_scale, out, accumulate, x2, zero_indptr)
        if accumulate:
            written = has                                            # modes 1 / 2 store only the rows that receive edges
        elif zero_indptr is not None:
            written = has | (zero_indptr[1:] == zero_indptr[:-1])[:n_rows]
        else:
            written = torch.ones(n_rows, dtype=torch.bool)
        self._mirror(wire, written, res)
        return res

    def _aggregate(self, x, index, reduce_op, n_rows, y=None, message_op="add", src_scale=None, dst_scale=None, out=None,
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
