// vtk_decl.h -- NOT VTK.  Hand-written, minimal DECLARATIONS of the dozen VTK classes and macros that
// vtk/vtkCudaReconstructionFilter.{h,cxx} name, written from the VTK API's public documentation for one purpose: letting
// `g++ -fsyntax-only` parse and type-check those two files in an image that has no VTK (tests/test_vtk_syntax.py).
// Nothing here is implemented, nothing is linked, nothing runs.  A green syntax check proves that the binding's C++ is
// well-formed against THESE declarations (names, arities, argument types as the real classes have them); it is NOT
// evidence that the binding builds or behaves against a real VTK.  vtk/README.md says the same.
#ifndef DMI_TEST_VTK_DECL_H
#define DMI_TEST_VTK_DECL_H

#include <cstring>
#include <iostream>
#include <ostream>

using std::ostream;
typedef long long vtkIdType;

class vtkIndent {};
inline ostream& operator<<(ostream& os, const vtkIndent&) { return os; }

class vtkObjectBase {
public:
  virtual ~vtkObjectBase();
  virtual void Delete();
  void Register(vtkObjectBase*);
  void UnRegister(vtkObjectBase*);
};

class vtkObject : public vtkObjectBase {
public:
  static vtkObject* New();
  virtual void Modified();
  virtual void PrintSelf(ostream& os, vtkIndent indent);

protected:
  vtkObject();
};

// the type macro: a Superclass typedef and the class-name / down-cast members the pipeline relies on
#define vtkTypeMacro(thisClass, superclass)                   \
  typedef superclass Superclass;                              \
  virtual const char* GetClassName() const;                   \
  static thisClass* SafeDownCast(vtkObjectBase* o);

#define vtkSetMacro(name, type)        \
  virtual void Set##name(type _arg) {  \
    if (this->name != _arg) {          \
      this->name = _arg;               \
      this->Modified();                \
    }                                  \
  }
#define vtkGetMacro(name, type) \
  virtual type Get##name() { return this->name; }
// owns a heap copy of the string (released by Set...(nullptr))
#define vtkSetStringMacro(name)                                   \
  virtual void Set##name(const char* _arg) {                      \
    if (this->name == nullptr && _arg == nullptr) return;         \
    delete[] this->name;                                          \
    this->name = nullptr;                                         \
    if (_arg) {                                                   \
      this->name = new char[std::strlen(_arg) + 1];               \
      std::strcpy(this->name, _arg);                              \
    }                                                             \
    this->Modified();                                             \
  }
#define vtkStandardNewMacro(thisClass) \
  thisClass* thisClass::New() { return new thisClass; }
// reference-counted object member
#define vtkCxxSetObjectMacro(cls, name, type)  \
  void cls::Set##name(type* _arg) {            \
    if (this->name == _arg) return;            \
    type* old = this->name;                    \
    this->name = _arg;                         \
    if (_arg) _arg->Register(this);            \
    if (old) old->UnRegister(this);            \
    this->Modified();                          \
  }
#define vtkErrorMacro(x)              \
  do {                                \
    std::cerr << "ERROR: " x << "\n"; \
  } while (false)

class vtkInformationIntegerVectorKey;
class vtkInformationStringKey;

class vtkInformation : public vtkObject {
public:
  void Set(vtkInformationStringKey* key, const char* value);
  void Set(vtkInformationIntegerVectorKey* key, const int* values, int length);
  void Get(vtkInformationIntegerVectorKey* key, int* values);
};

class vtkInformationVector : public vtkObject {
public:
  vtkInformation* GetInformationObject(int index);
};

class vtkStreamingDemandDrivenPipeline {
public:
  static vtkInformationIntegerVectorKey* WHOLE_EXTENT();
};

class vtkAlgorithm : public vtkObject {
public:
  static vtkInformationStringKey* INPUT_REQUIRED_DATA_TYPE();
  virtual void SetNumberOfInputPorts(int n);
  virtual void SetNumberOfOutputPorts(int n);
};

class vtkImageAlgorithm : public vtkAlgorithm {
public:
  vtkTypeMacro(vtkImageAlgorithm, vtkAlgorithm);
  void PrintSelf(ostream& os, vtkIndent indent) override;

protected:
  vtkImageAlgorithm();
  ~vtkImageAlgorithm() override;
  virtual int RequestData(vtkInformation*, vtkInformationVector**, vtkInformationVector*);
  virtual int RequestInformation(vtkInformation*, vtkInformationVector**, vtkInformationVector*);
  virtual int RequestUpdateExtent(vtkInformation*, vtkInformationVector**, vtkInformationVector*);
  virtual int FillInputPortInformation(int port, vtkInformation* info);
};

class vtkMatrix4x4 : public vtkObject {
public:
  double GetElement(int i, int j) const;
};

class vtkAbstractArray : public vtkObject {
public:
  virtual void SetName(const char* name);
  virtual void SetNumberOfComponents(int n);
  virtual void SetNumberOfTuples(vtkIdType n);
  vtkIdType GetNumberOfTuples() const;
};

class vtkDataArray : public vtkAbstractArray {
public:
  virtual void FillComponent(int component, double value);
};

class vtkDoubleArray : public vtkDataArray {
public:
  static vtkDoubleArray* New();
  static vtkDoubleArray* SafeDownCast(vtkObjectBase* o);
  double* GetPointer(vtkIdType id);
};

class vtkFieldData : public vtkObject {
public:
  int AddArray(vtkAbstractArray* array);
  vtkDataArray* GetArray(const char* name);
};
class vtkDataSetAttributes : public vtkFieldData {};
class vtkCellData : public vtkDataSetAttributes {};
class vtkPointData : public vtkDataSetAttributes {};

class vtkDataObject : public vtkObject {
public:
  virtual void ShallowCopy(vtkDataObject* src);
};

class vtkImageData : public vtkDataObject {
public:
  static vtkImageData* GetData(vtkInformationVector* v, int i = 0);
  virtual int* GetDimensions();
  virtual void GetDimensions(int dims[3]);
  virtual void GetOrigin(double origin[3]);
  virtual void GetSpacing(double spacing[3]);
  vtkIdType GetNumberOfCells();
  vtkCellData* GetCellData();
  vtkPointData* GetPointData();
};

// a smart pointer that creates its object
template <class T>
class vtkNew {
public:
  vtkNew() : Object(T::New()) {}
  ~vtkNew() {
    if (this->Object) this->Object->Delete();
  }
  T* operator->() const { return this->Object; }
  T* Get() const { return this->Object; }
  T* GetPointer() const { return this->Object; }
  operator T*() const { return this->Object; }

private:
  vtkNew(const vtkNew&) = delete;
  void operator=(const vtkNew&) = delete;
  T* Object;
};

#endif
