"""Number formatting of the report (R/syops/utils.py:10-46: same thresholds, units and rounding)."""


def _scaled(value, units, precision, table, default_suffix):
    if units is None:
        for div, suffix in table:
            if value // div > 0:
                return str(round(value / float(div), precision)) + ' ' + suffix
        return str(value) + default_suffix
    for div, suffix in table:
        if units == suffix:
            return str(round(value / float(div), precision)) + ' ' + units
    return str(value) + default_suffix


def syops_to_string(syops, units=None, precision=2):
    return _scaled(syops, units, precision, ((10 ** 9, 'G Ops'), (10 ** 6, 'M Ops'), (10 ** 3, 'K Ops')), ' Ops')


def params_to_string(params_num, units=None, precision=2):
    if units is None:
        if params_num // 10 ** 6 > 0:
            return str(round(params_num / 10 ** 6, precision)) + ' M'
        if params_num // 10 ** 3:
            return str(round(params_num / 10 ** 3, precision)) + ' k'
        return str(params_num)
    if units == 'M':
        return str(round(params_num / 10. ** 6, precision)) + ' ' + units
    if units == 'K':
        return str(round(params_num / 10. ** 3, precision)) + ' ' + units
    return str(params_num)
