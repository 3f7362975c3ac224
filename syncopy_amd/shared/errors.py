"""Exception types on the compute-function boundary (names and argument order of
syncopy/shared/errors.py:22-140 so that callers can catch them unchanged)."""


class SPYError(Exception):
    pass


class SPYTypeError(SPYError):
    def __init__(self, var, varname="", expected=""):
        self.var, self.varname, self.expected = var, varname, expected

    def __str__(self):
        msg = "Wrong type{vn}{ex}{act}"
        return msg.format(
            vn=f" of `{self.varname}`:" if self.varname else ":",
            ex=f" expected {self.expected}" if self.expected else "",
            act=f" found {type(self.var).__name__}",
        )


class SPYValueError(SPYError):
    def __init__(self, legal, varname="", actual=""):
        self.legal, self.varname, self.actual = legal, varname, actual

    def __str__(self):
        return "Invalid value{vn} {act}; expected {leg}".format(
            vn=f" of `{self.varname}`:" if self.varname else ":",
            act=f"'{self.actual}'" if self.actual != "" else "",
            leg=self.legal,
        )


class SPYParallelError(SPYError):
    pass


class SPYIOError(SPYError):
    pass


def SPYWarning(msg, caller=None):
    import warnings
    warnings.warn(f"Syncopy{(' <' + caller + '>') if caller else ''} WARNING: {msg}", stacklevel=3)


def SPYInfo(msg, caller=None):
    import logging
    logging.getLogger("syncopy_amd").info(msg)
