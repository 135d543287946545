"""paddle.static.Variable: only referenced in isinstance checks (there is no static graph mode here)."""


class Variable(object):
    pass
