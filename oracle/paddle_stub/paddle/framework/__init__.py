class _Core:
    class VarDesc:
        class VarType:
            FP16, FP32, FP64, INT32, INT64, BOOL = "float16", "float32", "float64", "int32", "int64", "bool"

    @staticmethod
    def get_cuda_current_device_id():
        return 0

    @staticmethod
    def is_compiled_with_cuda():
        return False


core = _Core()


def in_dygraph_mode():
    return True


def _non_static_mode():
    return True
